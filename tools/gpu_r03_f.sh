#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -m gpu -q -p no:cacheprovider -k "collect or workload or c1 or device_bam" > gpurun_out/r03f_pytest.txt 2>&1
tail -5 gpurun_out/r03f_pytest.txt
timeout 600 python tools/scan_ab.py ring4=svim_amd/libsvx.so ring0=svim_amd/variants/libsvx_ring0.so ring8=svim_amd/variants/libsvx_ring8.so ring2=svim_amd/variants/libsvx_ring2.so ring4b=svim_amd/libsvx.so ring0b=svim_amd/variants/libsvx_ring0.so > gpurun_out/r03f_scan_ab.txt 2>&1
cat gpurun_out/r03f_scan_ab.txt
timeout 600 python tools/device_reader_rate.py 180000 2048 > gpurun_out/r03f_device_reader_rate.txt 2>&1
grep -v "amdgpu.ids\|bamio pass\|   pass\|bamio 64" gpurun_out/r03f_device_reader_rate.txt
