#!/bin/bash
# Round 4, last pass after the 4-word class moved to the staircase kernel: whole suite, then the evidence that depends on the edit kernels (bench lines, kernel stats,
# timeline, the two SQ PMC passes, class profile); the reader / inflater / small-batch evidence of tools/gpu_evidence_r04.sh is unaffected and stays.
tag=r04p
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/${tag}_pytest.txt 2> gpurun_out/${tag}_pytest.err
tail -9 gpurun_out/${tag}_pytest.txt
tag=r04
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r04_bench_c1.json')); k=d["kernels"]; e=d["end_to_end"]
print("ms/step %.2f value %.4g edit %.2f scan %.3f first %.1f e2e %.4g"%(d["ms_per_step"], d["value"], k["cluster_edit_distance_ms"], k["k_cigar_scan_ms"], d["first_step_ms"], e["bam_file_reads_per_s"]))
P
cd /tmp
B="--steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end"
rm -rf /tmp/kt && (cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py $B > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null
python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline.txt
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
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
python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_small_batch_latency.txt
ls gpurun_out/${tag}_* | wc -l
