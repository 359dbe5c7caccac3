#!/bin/bash
# Round-6 GPU runs, one parameterised runner:   gpurun --timeout N -- 'bash tools/gpu_r06.sh <step> [tag]'
# Everything a step writes goes to gpurun_out/<tag>_*; what is to be judged is copied to profiles/ by hand afterwards.
set -u
step=${1:-first}; tag=${2:-r06}
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$PWD
case "$step" in
micro)
  for b in ${BINS:-valu_banks}; do timeout 300 tools/micro/$b.bin > $out/${tag}_$b.txt 2>&1; echo "$b rc=$?"; done
  cat $out/${tag}_valu_banks.txt 2>/dev/null | head -80
  ;;
suite)
  timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > $out/${tag}_pytest.txt 2>&1
  echo "suite: rc=$? $(tail -1 $out/${tag}_pytest.txt)"
  ;;
bench)
  python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; cut -c1-3000 $out/${tag}_bench.json
  ;;
ab)
  # interleaved A/B of library variants (tools/build_variants.sh):  VARIANTS="tree a b" WL="c1"
  bash tools/ab.sh "${WL:-c1}" ${VARIANTS:-tree} 2>&1 | tee $out/${tag}_ab.txt
  ;;
serial)
  # stand-alone duration of every edit launch (SVX_EDIT_SERIAL) and the per-class profile of the rounds (SVX_EDIT_PROFILE): stderr lines of one step
  SVX_EDIT_SERIAL=1 SVX_EDIT_PROFILE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --workload ${WL:-c1} 2>&1 >/dev/null | grep "^{" > $out/${tag}_edit_class_profile.jsonl
  grep edit_launch $out/${tag}_edit_class_profile.jsonl | tail -8
  ;;
timeline)
  rm -rf /tmp/kt && (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --workload ${WL:-c1} > /dev/null 2> /tmp/kt.err)
  db=$(find /tmp/kt -name "*.db" | head -1)
  python tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv > /dev/null
  python tools/rocpd_timeline.py $db > $out/${tag}_step_timeline.txt
  grep "k_edit\|k_cigar\|k_cluster" $out/${tag}_step_timeline.txt
  ;;
*) echo "unknown step $step"; exit 2;;
esac
