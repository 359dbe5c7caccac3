#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "edge_cases or edit_distance or per_read or cluster or golden or device_bam" > gpurun_out/r03q_pytest.txt 2>&1
tail -25 gpurun_out/r03q_pytest.txt
timeout 600 python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03q_small_batch_latency.txt
