#!/bin/bash
# round 3, call b: the rewritten DEFLATE decoder on the GPU (tests vs zlib, rate, per-symbol cost)
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "bgzf or reader_with_gpu or long_cigar or c1_config0 or bam_pipeline" > gpurun_out/r03b_pytest.txt 2>&1
tail -15 gpurun_out/r03b_pytest.txt
timeout 300 python tools/bgzf_inflate_rate.py 60000 > gpurun_out/r03b_bgzf_inflate_rate.txt 2>&1
cat gpurun_out/r03b_bgzf_inflate_rate.txt
timeout 300 python tools/bgzf_symbol_cost.py > gpurun_out/r03b_bgzf_symbol_cost.txt 2>&1
cat gpurun_out/r03b_bgzf_symbol_cost.txt
