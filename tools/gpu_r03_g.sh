#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "device_bam or bam_pipeline or c1_config0 or bam_path_native or dropin_collect_to_cluster or long_cigar or bench_harness_on_a_bam" > gpurun_out/r03g_pytest.txt 2>&1
tail -5 gpurun_out/r03g_pytest.txt
timeout 600 python tools/device_reader_rate.py 180000 2048 > gpurun_out/r03g_device_reader_rate.txt 2>&1
grep -v "amdgpu.ids\|bamio pass\|   pass\|bamio 64" gpurun_out/r03g_device_reader_rate.txt
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r03g_bench_c1.json 2> gpurun_out/r03g_bench_c1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03g_bench_c1.json'))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["first_step_ms"])
print(json.dumps(d.get("end_to_end"), indent=0)[:3500])
PY
