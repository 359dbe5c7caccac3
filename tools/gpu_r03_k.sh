#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "device_bam or bam_pipeline or bgzf or c1_config0_through" > gpurun_out/r03k_pytest.txt 2>&1
tail -3 gpurun_out/r03k_pytest.txt
timeout 900 python tools/device_reader_rate.py 180000 8192 > gpurun_out/r03k_device_reader_rate.txt 2>&1
grep -v "amdgpu.ids\|bamio pass\|   pass\|bamio 64" gpurun_out/r03k_device_reader_rate.txt
timeout 300 python tools/bgzf_symbol_cost.py 2>&1 | grep stored
