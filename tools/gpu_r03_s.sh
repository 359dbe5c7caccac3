#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline_edit']['seconds'], d['roofline_edit']['frac_issue_cycles'])"
SVX_EDIT_NO_EARLY=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no split:', d['value'], d['ms_per_step'], d['roofline_edit']['seconds'])"
python bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "edit or workload or c1 or cluster" > gpurun_out/r03s_pytest.txt 2>&1; tail -3 gpurun_out/r03s_pytest.txt
