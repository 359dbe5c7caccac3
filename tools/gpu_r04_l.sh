#!/bin/bash
# Round 4, last pass: whole suite on the final code (multi-window inflate steps, slot-lifetime test), then the evidence pass for profiles/.
tag=r04l
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/${tag}_pytest.txt 2> gpurun_out/${tag}_pytest.err
tail -8 gpurun_out/${tag}_pytest.txt; grep -c "hipHostUnregister" gpurun_out/${tag}_pytest.txt gpurun_out/${tag}_pytest.err
bash tools/gpu_evidence_r04.sh r04 > gpurun_out/${tag}_evidence.log 2>&1
tail -5 gpurun_out/${tag}_evidence.log
