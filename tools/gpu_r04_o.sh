#!/bin/bash
# Round 4: the 4-word band class as a (narrowing) staircase class (SVX_STAIR4) against dropping it (SVX_NO_SLIDE4) and against the sliding window it was.
tag=r04o
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
for v in default noslide4 stair4 stair4_qmin2 default noslide4 stair4; do
  if [ $v = default ]; then timeout 300 python bench.py $B > gpurun_out/${tag}_bench_$v.json 2> gpurun_out/${tag}_bench_$v.err
  else SVX_LIB=svim_amd/variants/libsvx_$v.so timeout 300 python bench.py $B > gpurun_out/${tag}_bench_$v.json 2> gpurun_out/${tag}_bench_$v.err; fi
  python - $v <<'P'
import json,sys
name=sys.argv[1]
try:
    d=json.load(open('gpurun_out/r04o_bench_%s.json'%name))
    k=d["kernels"]; print("%-16s"%name, "ms/step %.2f edit %.2f wc_issued %.3g wc_band %.3g frac_issue %.3f"%(d["ms_per_step"], k["cluster_edit_distance_ms"], k["edit_wordcols_issued"], k["edit_wordcols_band_kernels"], d["roofline_edit"]["frac_issue_cycles"]))
except Exception as e: print(name, "failed", e)
P
done
for v in stair4 stair4_qmin2; do
SVX_LIB=svim_amd/variants/libsvx_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "edit_distance or cluster_scheduling" > gpurun_out/${tag}_pytest_edit_$v.txt 2>&1
tail -2 gpurun_out/${tag}_pytest_edit_$v.txt
done
