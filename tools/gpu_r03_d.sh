#!/bin/bash
# round 3, call d: device BAM decode tests; inflate LDS variants (ring 2 KiB -> 27 waves / CU, + 9-bit table -> 32 waves / CU)
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "device_bam or bam_pipeline or c1_config0_through or bam_path_native or dropin_collect_to_cluster or long_cigar or bench_harness_on_a_bam" > gpurun_out/r03d_pytest.txt 2>&1
tail -30 gpurun_out/r03d_pytest.txt
for v in ; do
  lib=$PWD/svim_amd/variants/libsvx_$v.so; [ "$v" = "tree" ] && lib=$PWD/svim_amd/libsvx.so
  echo "== $v" >> gpurun_out/r03d_bgzf_variants.txt
  SVX_LIB=$lib timeout 300 python tools/bgzf_inflate_rate.py 60000 2>&1 | grep "GPU," >> gpurun_out/r03d_bgzf_variants.txt
done
cat gpurun_out/r03d_bgzf_variants.txt
SVX_BAM_TIMING=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03d_bench_c1.json 2> gpurun_out/r03d_bench_c1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d_bench_c1.json'))
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("end_to_end"), indent=0)[:3000])
PY
grep -i "bamio" gpurun_out/r03d_bench_c1.err | tail -12
