#!/bin/bash
# Round 4, third GPU pass: prep with one shifted alignment, v_bfe_i32 masks, stream priorities A/B, whole suite (reader fix, foreign BAMs, configs[3] stand-in).
tag=r04c
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "edit_distance or cluster_scheduling" > gpurun_out/${tag}_pytest_edit.txt 2>&1
tail -3 gpurun_out/${tag}_pytest_edit.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
for prio in band equal full; do
  SVX_EDIT_PRIO=$prio timeout 300 python bench.py $B > gpurun_out/${tag}_bench_prio_$prio.json 2> gpurun_out/${tag}_bench_prio_$prio.err
done
python - <<'P'
import json
for prio in ("band","equal","full"):
    try:
        d=json.load(open('gpurun_out/r04c_bench_prio_%s.json'%prio))
        k=d["kernels"]; print("prio",prio, "ms/step %.2f edit %.2f wc_issued %.3g wc_band %.3g frac_issue %.3f first %.1f"%(d["ms_per_step"], k["cluster_edit_distance_ms"], k["edit_wordcols_issued"], k["edit_wordcols_band_kernels"], d["roofline_edit"]["frac_issue_cycles"], d["first_step_ms"]))
    except Exception as e: print(prio, "failed", e)
P
cd /tmp; rm -rf /tmp/kt && (cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null && python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline.txt
cd $R
grep -E "k_edit|k_ins_pairs|k_class_bounds" gpurun_out/${tag}_step_timeline.txt | head -12
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > gpurun_out/${tag}_pytest.txt 2>&1
tail -22 gpurun_out/${tag}_pytest.txt
