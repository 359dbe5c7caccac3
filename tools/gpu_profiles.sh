#!/bin/bash
# Round-2 evidence for profiles/: kernel stats + timeline of the default bench, PMC passes (edit kernels: SQ counters; k_cigar_scan: FETCH_SIZE and
# WRITE_SIZE in separate passes), the per-class edit profile, bench lines of every workload.  Usage: tools/gpu_profiles.sh <tag>
tag=${1:-r02}
export TMPDIR=/tmp
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp
B="--steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end"
rm -rf /tmp/kt && (cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py $B > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null
python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline.txt
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
  t=$(echo $pass | tr ' ' '_' | cut -c1-24)
  rm -rf /tmp/pmc_$t
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$t -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/pmc_$t.err)
  db=$(find /tmp/pmc_$t -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db $R/gpurun_out/${tag}_pmc_$t.csv > /dev/null
done
cd $R
SVX_EDIT_SERIAL=1 SVX_EDIT_PROFILE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> gpurun_out/${tag}_edit_profile_raw.txt
grep -E "edit_profile|edit_launch|edit_guess" gpurun_out/${tag}_edit_profile_raw.txt > gpurun_out/${tag}_edit_class_profile.jsonl
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err
python bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline > gpurun_out/${tag}_bench_c2.json 2>/dev/null
for pmd in 1000 5000 20000 100000; do python bench.py --steps 5 --warmup 2 --workload c4 --partition-max-distance $pmd --no-cpu-baseline > gpurun_out/${tag}_bench_c4_pmd$pmd.json 2>/dev/null; done
SVX_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_c1_dist_path_1rank.json 2>/dev/null
bash tools/host_probe.sh > gpurun_out/${tag}_host_probe.txt 2>&1
[ -x tools/micro/read_bw.bin ] || hipcc --offload-arch=gfx950 -O3 -o tools/micro/read_bw.bin tools/micro/read_bw.hip
tools/micro/read_bw.bin > gpurun_out/${tag}_read_bw.txt 2>&1
# calibration of the FETCH_SIZE correction: the probe reads a known number of bytes per launch with the same 16 B / lane loads
rm -rf /tmp/pmc_cal; (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_cal -o p -- $R/tools/micro/read_bw.bin > /dev/null 2>&1)
db=$(find /tmp/pmc_cal -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_summary.py $db gpurun_out/${tag}_pmc_FETCH_SIZE_read_bw_calibration.csv > /dev/null
SVX_BAM_GPU_INFLATE=0 python tools/reader_scaling.py 60000 > gpurun_out/${tag}_reader_scaling.txt 2>&1
python tools/bgzf_inflate_rate.py 60000 > gpurun_out/${tag}_bgzf_inflate_rate.txt 2>&1
python tools/bgzf_symbol_cost.py > gpurun_out/${tag}_bgzf_symbol_cost.txt 2>&1
python tools/reader_gpu_inflate.py 240000 > gpurun_out/${tag}_reader_gpu_inflate.txt 2>&1
ls -la gpurun_out/${tag}_* | head -40
