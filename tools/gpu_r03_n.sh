#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "device_bam or bgzf" > gpurun_out/r03n_pytest.txt 2>&1
tail -3 gpurun_out/r03n_pytest.txt
timeout 300 python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids | tail -5
SVX_LIB=svim_amd/variants/libsvx_prof.so timeout 300 python tools/inflate_profile.py 60000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03n_inflate_profile.txt
timeout 900 python tools/device_reader_rate.py 180000 8192 2>&1 | grep "default" 
