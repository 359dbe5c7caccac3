#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "device_bam or bam_pipeline or c1_config0_through or bam_path_native or dropin_collect_to_cluster or long_cigar or bench_harness_on_a_bam or bgzf" > gpurun_out/r03e_pytest.txt 2>&1
tail -30 gpurun_out/r03e_pytest.txt
timeout 300 python tools/bgzf_inflate_rate.py 60000 2>&1 | grep "GPU," > gpurun_out/r03e_bgzf_inflate_rate.txt; cat gpurun_out/r03e_bgzf_inflate_rate.txt
timeout 900 python tools/device_reader_rate.py 180000 2048 1024 > gpurun_out/r03e_device_reader_rate.txt 2>&1
grep -v "amdgpu.ids\|bamio pass\|   pass" gpurun_out/r03e_device_reader_rate.txt
