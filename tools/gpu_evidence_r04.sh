#!/bin/bash
# Round-4 evidence for profiles/: the whole -m gpu suite, the bench lines of every workload, kernel stats + step timeline, PMC passes (k_cigar_scan traffic, edit kernels),
# the DEFLATE decoder (rate, per-symbol cost), the device-resident BAM reader (rates, stage times, kernel timeline), small-batch latency, host probe.
tag=${1:-r04}
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err
cd /tmp
B="--steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end"
rm -rf /tmp/kt && (cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py $B > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null
python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline.txt
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $pass | tr ' ' '_' | cut -c1-24)
  rm -rf /tmp/pmc_$t
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$t -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/pmc_$t.err)
  db=$(find /tmp/pmc_$t -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db $R/gpurun_out/${tag}_pmc_$t.csv > /dev/null
done
cd $R
SVX_EDIT_SERIAL=1 SVX_EDIT_PROFILE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> gpurun_out/${tag}_edit_profile_raw.txt
grep -E "edit_profile|edit_launch|edit_guess|edit_band_fit" gpurun_out/${tag}_edit_profile_raw.txt > gpurun_out/${tag}_edit_class_profile.jsonl; rm -f gpurun_out/${tag}_edit_profile_raw.txt
python bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline > gpurun_out/${tag}_bench_c2.json 2>/dev/null
for pmd in 1000 5000 20000 100000; do python bench.py --steps 5 --warmup 2 --workload c4 --partition-max-distance $pmd --no-cpu-baseline > gpurun_out/${tag}_bench_c4_pmd$pmd.json 2>/dev/null; done
SVX_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_c1_dist_path_1rank.json 2>/dev/null
port=29577
SVX_BENCH_BACKEND=gloo SVX_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 5 --warmup 2 --workload c2 --scale 0.25 --foreign-frac 0.5 --partition-max-distance 5000 > gpurun_out/${tag}_bench_c2_two_ranks_one_gpu_foreign.json 2>/dev/null
bash tools/host_probe.sh > gpurun_out/${tag}_host_probe.txt 2>&1
python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_inflate_rate.txt
[ -f svim_amd/variants/libsvx_prof.so ] && SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_inflate_profile.txt
[ -f svim_amd/variants/libsvx_prof.so ] && SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py 40000 qual 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_inflate_profile_qual.txt
python tools/bgzf_symbol_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_symbol_cost.txt
python tools/device_reader_rate.py 180000 8192 2>&1 | grep -v "amdgpu.ids\|bamio pass\|   pass\|bamio 64" > gpurun_out/${tag}_device_reader_rate.txt
cd /tmp; rm -rf /tmp/rt
SVX_READER_ONE=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/rt -o p -- python $R/tools/device_reader_rate.py 180000 8192 > /tmp/rt.out 2>&1
db=$(find /tmp/rt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/reader_timeline.py $db | tail -14 > $R/gpurun_out/${tag}_reader_timeline.txt
cd $R
python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_small_batch_latency.txt
cd /tmp; rm -rf /tmp/sb; rocprofv3 --kernel-trace -d /tmp/sb -o p -- python $R/tools/small_batch_trace.py run 1000 > /tmp/sb.out 2>&1
db=$(find /tmp/sb -name "*.db" | head -1); [ -n "$db" ] && (grep "last cluster" /tmp/sb.out; python $R/tools/small_batch_trace.py show $db) > $R/gpurun_out/${tag}_small_batch_trace.txt
cd $R
ls -la gpurun_out/${tag}_* | head -50
# the end_to_end block on a MILLION records with base qualities (27 GB of stream in 14 chunks; the default bench run uses 300 k to stay within minutes)
timeout 900 python bench.py --end-to-end-child --e2e-records-qual 1000000 --e2e-records 180000 > gpurun_out/${tag}_end_to_end_1M_records.json 2> gpurun_out/${tag}_end_to_end_1M_records.err || echo "1M-record end-to-end run did not finish (disk / time)"
