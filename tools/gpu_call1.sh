#!/bin/bash
# round-2 GPU call 1: parity of the reworked edit kernels, variant A/B, stand-alone launch durations, PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/c1_pytest.txt
cat gpurun_out/c1_pytest.txt
tools/edit_variants.sh c1 base v1 v2 tree 2>&1 | tee gpurun_out/c1_variants.txt
SVX_EDIT_SERIAL=1 SVX_EDIT_PROFILE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c1_serial.json 2> gpurun_out/c1_serial.err
grep -c edit_launch gpurun_out/c1_serial.err
SVX_LIB=$PWD/svim_amd/variants/libsvx_base.so SVX_EDIT_SERIAL=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c1_serial_base.json 2> gpurun_out/c1_serial_base.err
cd /tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_IFETCH"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  (cd $R && SVX_EDIT_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pmc_$tag.err)
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db $R/gpurun_out/c1_pmc_$tag.csv > /dev/null
  ls -la $db
done
