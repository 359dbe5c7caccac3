#!/bin/bash
# Round 4: full-matrix classes of 10 and 14 words per lane beside 12 and 16 (a pair pads up to 1/8 of its rows instead of 1/4): A/B against the build without them, then the whole suite.
tag=r04q
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
for v in default oldwide default oldwide; do
  if [ $v = default ]; then timeout 200 python bench.py $B > gpurun_out/${tag}_bench_$v.json 2> gpurun_out/${tag}_bench_$v.err
  else SVX_LIB=svim_amd/variants/libsvx_$v.so timeout 200 python bench.py $B > gpurun_out/${tag}_bench_$v.json 2> gpurun_out/${tag}_bench_$v.err; fi
  python - $v <<'P'
import json,sys
name=sys.argv[1]
try:
    d=json.load(open('gpurun_out/r04q_bench_%s.json'%name))
    k=d["kernels"]; print("%-10s"%name, "ms/step %.2f edit %.2f wc_issued %.4g wc_useful %.4g wc_band %.3g frac_issue %.3f"%(d["ms_per_step"], k["cluster_edit_distance_ms"], k["edit_wordcols_issued"], k["edit_wordcols_useful"], k["edit_wordcols_band_kernels"], d["roofline_edit"]["frac_issue_cycles"]))
except Exception as e: print(name, "failed", e)
P
done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=3 > gpurun_out/${tag}_pytest.txt 2> gpurun_out/${tag}_pytest.err
tail -7 gpurun_out/${tag}_pytest.txt
