#!/bin/bash
# Round 4: the whole -m gpu suite once more at the final code (the log under profiles/), stderr kept.
tag=r04m
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/${tag}_pytest.txt 2> gpurun_out/${tag}_pytest.err
tail -14 gpurun_out/${tag}_pytest.txt; grep -c "hipHostUnregister" gpurun_out/${tag}_pytest.txt gpurun_out/${tag}_pytest.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
