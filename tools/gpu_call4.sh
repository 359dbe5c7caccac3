#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
SVX_SKIP_SLOW=1 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -30 | tee gpurun_out/c4_pytest.txt
python bench.py --steps 5 --warmup 2 > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err
tail -c 3000 gpurun_out/c4_bench.err | grep -v "amdgpu.ids" | tail -20
python - <<'PY'
import json
for l in open("gpurun_out/c4_bench.json"):
    if l.startswith("{"):
        j = json.loads(l)
        print("ms/step", j["ms_per_step"], "value", j["value"])
        print("end_to_end", json.dumps(j.get("end_to_end"), indent=1)[:3000])
        print("cpu_baseline", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("parity_vs_gpu_on_sample"))
        print("roofline_edit", {k: v for k, v in (j.get("roofline_edit") or {}).items() if k not in ("note", "pmc", "kernels", "unit")})
PY
SVX_BAM_TIMING=1 python tools/reader_scaling.py 60000 2>&1 | tee gpurun_out/c4_reader_scaling.txt | tail -20
