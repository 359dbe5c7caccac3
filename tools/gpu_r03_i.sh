#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
R=$PWD
cd /tmp
rm -rf /tmp/rt
SVX_READER_ONE=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/rt -o p -- python $R/tools/device_reader_rate.py 180000 8192 > /tmp/rt.out 2>&1
tail -5 /tmp/rt.out
db=$(find /tmp/rt -name "*.db" | head -1)
python $R/tools/reader_timeline.py $db > $R/gpurun_out/r03i_reader_timeline.txt 2>&1
cat $R/gpurun_out/r03i_reader_timeline.txt | head -80
