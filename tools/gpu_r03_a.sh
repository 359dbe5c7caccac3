#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (new: combine consumers, writers, c1 full size, rank exchange, two ranks on one GPU) + the default bench line
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/r03a_pytest.txt 2>&1
tail -25 gpurun_out/r03a_pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a_bench_c1.json 2> gpurun_out/r03a_bench_c1.err
tail -c 1500 gpurun_out/r03a_bench_c1.json
timeout 300 python tools/bgzf_inflate_rate.py 60000 > gpurun_out/r03a_bgzf_inflate_rate.txt 2>&1
cat gpurun_out/r03a_bgzf_inflate_rate.txt
