#!/bin/bash
# Round 4, tenth GPU pass: early full matrices, which priority their stream (and the main stream) should have.
tag=r04j
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err; }
run off SVX_EDIT_EARLY_FULLS=0
run normal SVX_EDIT_EARLY_FULLS=1 SVX_EDIT_EARLY_PRIO=normal
run high SVX_EDIT_EARLY_FULLS=1 SVX_EDIT_EARLY_PRIO=high
run high_mainhigh SVX_EDIT_EARLY_FULLS=1 SVX_EDIT_EARLY_PRIO=high SVX_MAIN_PRIO=high
run low SVX_EDIT_EARLY_FULLS=1 SVX_EDIT_EARLY_PRIO=low
run normal_mainhigh SVX_EDIT_EARLY_FULLS=1 SVX_EDIT_EARLY_PRIO=normal SVX_MAIN_PRIO=high
python - <<'P'
import json
for name in ("off", "normal", "high", "high_mainhigh", "low", "normal_mainhigh"):
    try:
        d=json.load(open('gpurun_out/r04j_bench_%s.json'%name))
        k=d["kernels"]; print("%-16s"%name, "ms/step %.2f edit %.2f frac_issue %.3f first %.1f"%(d["ms_per_step"], k["cluster_edit_distance_ms"], d["roofline_edit"]["frac_issue_cycles"], d["first_step_ms"]))
    except Exception as e: print(name, "failed", e)
P
for name in normal high_mainhigh; do
  case $name in normal) E="SVX_EDIT_EARLY_PRIO=normal";; high_mainhigh) E="SVX_EDIT_EARLY_PRIO=high SVX_MAIN_PRIO=high";; esac
  cd /tmp; rm -rf /tmp/kt && (cd $R && env SVX_EDIT_EARLY_FULLS=1 $E timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/kt.err)
  db=$(find /tmp/kt -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline_$name.txt
  cd $R
  echo "--- $name"; grep -E "k_edit|k_ins_pairs|k_class_bounds" gpurun_out/${tag}_step_timeline_$name.txt | head -20
done
