#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "edit_distance or per_read or cluster or golden" > gpurun_out/r03p_pytest.txt 2>&1
tail -5 gpurun_out/r03p_pytest.txt
timeout 600 python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03p_small_batch_latency.txt
SVX_EDIT_FEW_PAIRS=0 timeout 600 python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids | grep " reads (" 
