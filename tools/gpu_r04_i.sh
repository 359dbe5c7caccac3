#!/bin/bash
# Round 4, ninth GPU pass: early full matrices (SVX_EDIT_EARLY_FULLS) A/B + the tests that cover them; what hipMalloc costs (first step of a context).
tag=r04i
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "edit_distance or cluster_scheduling" > gpurun_out/${tag}_pytest_edit.txt 2>&1
tail -3 gpurun_out/${tag}_pytest_edit.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
for ef in 0 1; do
  SVX_EDIT_EARLY_FULLS=$ef timeout 300 python bench.py $B > gpurun_out/${tag}_bench_early_$ef.json 2> gpurun_out/${tag}_bench_early_$ef.err
done
python - <<'P'
import json
for ef in (0, 1):
    try:
        d=json.load(open('gpurun_out/r04i_bench_early_%d.json'%ef))
        k=d["kernels"]; print("SVX_EDIT_EARLY_FULLS=%d"%ef, "ms/step %.2f edit %.2f wc_issued %.3g wc_band %.3g frac_issue %.3f first %.1f"%(d["ms_per_step"], k["cluster_edit_distance_ms"], k["edit_wordcols_issued"], k["edit_wordcols_band_kernels"], d["roofline_edit"]["frac_issue_cycles"], d["first_step_ms"]))
    except Exception as e: print(ef, "failed", e)
P
tail -3 gpurun_out/${tag}_bench_early_1.err
cd /tmp; rm -rf /tmp/kt && (cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null && python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline.txt
cd $R
grep -E "k_edit|k_ins_pairs|k_class_bounds|k_cluster" gpurun_out/${tag}_step_timeline.txt | head -24
timeout 900 python -m pytest tests/test_gpu_workloads.py -m gpu -q -x -p no:cacheprovider --durations=5 > gpurun_out/${tag}_pytest_workloads.txt 2>&1
tail -9 gpurun_out/${tag}_pytest_workloads.txt
timeout 120 tools/micro/malloc_cost.bin > gpurun_out/${tag}_malloc_cost.txt 2>&1; cat gpurun_out/${tag}_malloc_cost.txt
