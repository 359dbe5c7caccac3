#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
for m in 1 2 3 1 2; do
SVX_EDIT_PARTS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('parts $m c1', d['ms_per_step'], d['roofline_edit']['seconds'])"
done
for m in 1 2 3; do
SVX_EDIT_PARTS=$m python bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('parts $m c2', d['ms_per_step'])"
done
