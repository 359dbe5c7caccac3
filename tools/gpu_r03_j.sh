#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "device_bam or bam_pipeline or bgzf or reader_with_gpu" > gpurun_out/r03j_pytest.txt 2>&1
tail -3 gpurun_out/r03j_pytest.txt
timeout 600 python tools/device_reader_rate.py 180000 8192 > gpurun_out/r03j_device_reader_rate.txt 2>&1
grep -v "amdgpu.ids\|bamio pass\|   pass\|bamio 64" gpurun_out/r03j_device_reader_rate.txt
R=$PWD; cd /tmp; rm -rf /tmp/rt
SVX_READER_ONE=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/rt -o p -- python $R/tools/device_reader_rate.py 180000 8192 > /tmp/rt.out 2>&1
db=$(find /tmp/rt -name "*.db" | head -1)
python $R/tools/reader_timeline.py $db > $R/gpurun_out/r03j_reader_timeline.txt 2>&1
tail -12 $R/gpurun_out/r03j_reader_timeline.txt
