#!/bin/bash
# GPU-side A/B of libsvx variants (tools/build_variants.sh): one short bench per variant, key numbers on one line each.
#   tools/edit_variants.sh out_prefix variant1 variant2 ...
out=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  lib=svim_amd/variants/libsvx_$v.so
  [ "$v" = "tree" ] && lib=svim_amd/libsvx.so
  SVX_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${out}_$v.json 2> gpurun_out/${out}_$v.err
  python - "$v" gpurun_out/${out}_$v.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(sys.argv[1], "ms/step %.2f" % j["ms_per_step"], "edit %.2f" % k["cluster_edit_distance_ms"], "linkage %.2f" % k["cluster_linkage_ms"],
          "scan %.3f" % k["k_cigar_scan_ms"], "wc_issued %s retry %s band %s guess %s" % (k.get("edit_wordcols_issued"), k.get("edit_wordcols_retry_rounds"),
          k.get("edit_wordcols_band_kernels"), k.get("edit_guess")), "clusters", j["counts"]["clusters"], "sigs", j["counts"]["signatures"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
