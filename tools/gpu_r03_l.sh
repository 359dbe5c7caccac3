#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "device_bam or bgzf" > gpurun_out/r03l_pytest.txt 2>&1
tail -3 gpurun_out/r03l_pytest.txt
timeout 300 python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python tools/bgzf_symbol_cost.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 900 python tools/device_reader_rate.py 180000 8192 > gpurun_out/r03l_device_reader_rate.txt 2>&1
grep -v "amdgpu.ids\|bamio pass\|   pass\|bamio 64" gpurun_out/r03l_device_reader_rate.txt
