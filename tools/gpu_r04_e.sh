#!/bin/bash
# Round 4, fifth GPU pass: why did the 8-rank configs[3] stand-in differ once?  (repeat it under switches), the first step of a fresh context.
tag=r04e
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
run_c3() {  # label, env...
  label=$1; shift
  port=$((29600 + RANDOM % 300))
  out=$(env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port tests/mp_c3_ranks_one_gpu.py /tmp/wg_$label.bam 0.003 3 2>/dev/null | grep -E "^C3_")
  echo "== $label: $out" | tee -a gpurun_out/${tag}_c3_repeat.txt
}
for i in 1 2 3 4; do run_c3 default_$i X=1; done
run_c3 host_reader SVX_BAM_DEVICE_DECODE=0
run_c3 host_reader2 SVX_BAM_DEVICE_DECODE=0
run_c3 no_narrow SVX_EDIT_NARROW=0
run_c3 no_shift SVX_EDIT_SHIFT_BOUNDS=0
run_c3 prio_band SVX_EDIT_PRIO=band
run_c3 few0 SVX_EDIT_FEW_PAIRS=0
run_c3 force_full SVX_EDIT_FORCE_FULL=1
SVX_ALLOC_STATS=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_first_step.json 2> gpurun_out/${tag}_bench_first_step.err
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_first_step.json')); print('first step', d['first_step_ms'], 'ms/step', d['ms_per_step'])"
tail -5 gpurun_out/${tag}_bench_first_step.err
cd /tmp; rm -rf /tmp/kt && (cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_first_step_timeline.txt
head -5 $R/gpurun_out/${tag}_first_step_timeline.txt
