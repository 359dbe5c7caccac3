#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "bgzf or reader_with_gpu" > gpurun_out/r03c_pytest.txt 2>&1
tail -3 gpurun_out/r03c_pytest.txt
for pad in 0 8192 24576; do
  echo "== SVX_INFLATE_LDS_PAD=$pad" >> gpurun_out/r03c_bgzf_inflate_rate.txt
  SVX_INFLATE_LDS_PAD=$pad timeout 300 python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03c_bgzf_inflate_rate.txt
done
cat gpurun_out/r03c_bgzf_inflate_rate.txt
timeout 300 python tools/bgzf_symbol_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03c_bgzf_symbol_cost.txt
cat gpurun_out/r03c_bgzf_symbol_cost.txt
