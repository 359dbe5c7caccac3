#!/bin/bash
# Round-6 GPU runs, one parameterised runner:   gpurun --timeout N -- 'bash tools/gpu_r06.sh <step> [tag]'
# Everything a step writes goes to gpurun_out/<tag>_*; what is to be judged is copied to profiles/ by hand afterwards.
set -u
step=${1:-first}; tag=${2:-r06}
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$PWD
case "$step" in
micro)
  for b in ${BINS:-valu_banks}; do timeout 300 tools/micro/$b.bin > $out/${tag}_$b.txt 2>&1; echo "$b rc=$?"; done
  cat $out/${tag}_valu_banks.txt 2>/dev/null | head -80
  ;;
quick)
  # the hot-path parity tests named by K (pytest -k), then an A/B:  K="edit_distance or c1_full" VARIANTS="tree tree@SVX_MAILBOX=0"
  timeout 1200 python -m pytest tests/ -x -q -m gpu -k "${K:-edit_distance}" --durations=5 > $out/${tag}_pytest_quick.txt 2>&1
  echo "quick: rc=$? $(tail -1 $out/${tag}_pytest_quick.txt)"
  [ -n "${VARIANTS:-}" ] && bash tools/ab.sh "${WL:-c1}" $VARIANTS 2>&1 | tee $out/${tag}_ab.txt
  [ -n "${E2E:-}" ] && bash tools/gpu_r06.sh e2e $tag
  ;;
suite)
  timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > $out/${tag}_pytest.txt 2>&1
  echo "suite: rc=$? $(tail -1 $out/${tag}_pytest.txt)"
  ;;
bench)
  python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; cut -c1-3000 $out/${tag}_bench.json
  ;;
ab)
  # interleaved A/B of library variants (tools/build_variants.sh):  VARIANTS="tree a b" WL="c1"
  bash tools/ab.sh "${WL:-c1}" ${VARIANTS:-tree} 2>&1 | tee $out/${tag}_ab.txt
  ;;
serial)
  # stand-alone duration of every edit launch (SVX_EDIT_SERIAL) and the per-class profile of the rounds (SVX_EDIT_PROFILE): stderr lines of one step
  SVX_EDIT_SERIAL=1 SVX_EDIT_PROFILE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --workload ${WL:-c1} 2>&1 >/dev/null | grep "^{" > $out/${tag}_edit_class_profile.jsonl
  grep edit_launch $out/${tag}_edit_class_profile.jsonl | tail -8
  ;;
timeline)
  KT=/tmp/kt_$$; rm -rf $KT; (cd /tmp && rocprofv3 --kernel-trace --stats -d $KT -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --workload ${WL:-c1} > /dev/null 2> $KT.err)
  db=$(find $KT -name "*.db" | head -1)
  python tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv > /dev/null
  python tools/rocpd_timeline.py $db > $out/${tag}_step_timeline.txt
  ALL=1 python tools/rocpd_timeline.py $db > $out/${tag}_step_timeline_all_kernels.txt
  grep "k_edit\|k_cigar\|k_cluster" $out/${tag}_step_timeline.txt
  ;;
strong)
  # the strong-scaling line on one rank and on 2 / 4 ranks that share this box's one GPU (gloo transport), and the driver's default (weak) line on 2 ranks with its strong_scaling block
  python bench.py --scaling strong --steps 5 --warmup 2 > $out/${tag}_bench_strong_1rank.json 2> $out/${tag}_bench_strong_1rank.err; echo "strong 1 rank rc=$?"
  for n in ${RANKS:-2 4}; do
    SVX_BENCH_BACKEND=gloo SVX_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n --scaling strong --steps 5 --warmup 2 > $out/${tag}_bench_strong_${n}ranks_one_gpu.json 2> $out/${tag}_bench_strong_${n}ranks_one_gpu.err; echo "strong $n ranks rc=$?"
  done
  SVX_BENCH_BACKEND=gloo SVX_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29588 bench.py --gpus 2 --steps 3 --warmup 1 --reads 200000 --contig-len 50000000 > $out/${tag}_bench_default_2ranks_one_gpu.json 2> $out/${tag}_bench_default_2ranks_one_gpu.err; echo "default line, 2 ranks rc=$?"
  python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/%s_bench_strong_*.json" % "'$tag'".strip("'"))) + sorted(glob.glob("gpurun_out/%s_bench_default_2ranks*.json" % "'$tag'".strip("'"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line:", e); continue
    sb = j.get("strong_scaling")
    print(f.split("/")[-1], "value %.3g reads/s, %.2f ms/step, clusters %s" % (j["value"], j["ms_per_step"], j["counts"].get("clusters_gathered", j["counts"].get("clusters"))), (j.get("ownership") or {}).get("max_over_mean_signatures"), (j.get("ownership") or {}).get("cuts"))
    if sb: print("   strong_scaling block:", {k: sb.get(k) for k in ("value", "ms_per_step", "n_gpus", "error")}, (sb.get("ownership") or {}).get("max_over_mean_signatures"))
PY
  ;;
inflate)
  # the two BGZF inflaters on the sample files of tools/bgzf_inflate_rate.py: lane per block (default) against wave per block (SVX_INFLATE_LANES=0)
  for m in 1 0; do SVX_INFLATE_LANES=$m timeout 900 python tools/bgzf_inflate_rate.py ${NREC:-60000} > $out/${tag}_bgzf_inflate_rate_lanes$m.txt 2>&1; echo "lanes=$m rc=$?"; grep "GPU\|identical\|Error\|error\|assert" $out/${tag}_bgzf_inflate_rate_lanes$m.txt; done
  ;;
readers)
  timeout 1500 python -m pytest tests/ -x -q -m gpu -k "bam or reader or inflate or bgzf" --durations=5 > $out/${tag}_pytest_readers.txt 2>&1
  echo "readers: rc=$? $(tail -1 $out/${tag}_pytest_readers.txt)"
  ;;
e2e)
  # the end_to_end block of bench.py alone (its child process), under the settings given:  E2E="name:ENV=V,ENV=V[:extra bench flags] ..."
  for spec in ${E2E:-default:}; do
    name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; flags=""; [ "$rest" != "$envs" ] && flags=${rest#*:}
    env $(echo $envs | tr ',' ' ') python bench.py --end-to-end-child --resident 6.5e7 $(echo $flags | tr ',' ' ') > $out/${tag}_e2e_$name.json 2> $out/${tag}_e2e_$name.err; echo "e2e $name rc=$?"
    python - $out/${tag}_e2e_$name.json $name <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    b = j["bam_file_with_base_qualities"]
    print("  %-12s warm %.3f M/s (%.3f s) cold %.3f M/s objects %.3f M/s | blocks gpu %d host %d kernel %.0f ms | no-qual %.2f M/s | host arrays %.2f M/s, library-pinned %.2f M/s" % (sys.argv[2], b["reads_per_s"] / 1e6, b["wall_s"],
          j["bam_file_first_pass_reads_per_s"] / 1e6, j["objects_materialised_reads_per_s"] / 1e6, b["inflate_blocks_gpu"], b["inflate_blocks_host_cores"], b["inflate_kernel_ms"],
          j["bam_file_without_base_qualities"]["reads_per_s"] / 1e6, j.get("host_arrays_reads_per_s", 0) / 1e6, j.get("host_arrays_library_pinned_reads_per_s", 0) / 1e6))
except Exception as e:
    print("  no line:", e)
PY
  done
  ;;
evidence)
  # what profiles/ holds at the end of the round, all at the SAME code: the default bench line, kernel stats + step timelines (rocprofv3 --kernel-trace --stats),
  # the four PMC passes (separate runs, no trace domains beside --kernel-trace), the stand-in workloads, small-batch latency, strong-scaling lines (one GPU, gloo)
  python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_c1.json 2> $out/${tag}_bench_c1.err; echo "bench c1 rc=$?"
  B="--steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end"
  KT=/tmp/kt_$$; rm -rf $KT; (cd /tmp && rocprofv3 --kernel-trace --stats -d $KT -o p -- python $R/bench.py $B > /dev/null 2> $KT.err)
  db=$(find $KT -name "*.db" | head -1)
  python tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv > /dev/null
  python tools/rocpd_timeline.py $db > $out/${tag}_step_timeline.txt
  ALL=1 python tools/rocpd_timeline.py $db > $out/${tag}_step_timeline_all_kernels.txt
  for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    t=$(echo $pass | tr ' ' '_' | cut -c1-24)
    P=/tmp/pmc_${t}_$$; rm -rf $P
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $P -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $P.err)
    db=$(find $P -name "*.db" | head -1)
    [ -n "$db" ] && python tools/pmc_summary.py $db $out/${tag}_pmc_$t.csv > /dev/null
  done
  python bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline > $out/${tag}_bench_c2.json 2>/dev/null
  python bench.py --steps 5 --warmup 2 --workload c4 --no-cpu-baseline > $out/${tag}_bench_c4.json 2>/dev/null
  python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_small_batch_latency.txt
  git rev-parse HEAD > $out/${tag}_evidence_commit.txt 2>/dev/null || true
  ls -la $out/${tag}_* | head -40
  ;;
rccl20)
  # the review's item: the single-rank RCCL test 20 x in fresh processes under NCCL_DEBUG=INFO with a 120 s watchdog - does process-group creation ever stall?
  ok=0; bad=0
  for k in $(seq 1 ${REPEAT:-20}); do
    t0=$(date +%s.%N)
    NCCL_DEBUG=INFO timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_multigpu_step_single_rank_device_path or test_bench_under_torchrun_multi_gpu_code_path_single_rank" > $out/${tag}_rccl_run.log 2>&1
    rc=$?; t1=$(date +%s.%N)
    echo "run $k: rc=$rc $(python -c "print('%.1f s' % ($t1 - $t0))") $(tail -1 $out/${tag}_rccl_run.log) | NCCL lines: $(grep -c NCCL $out/${tag}_rccl_run.log)" | tee -a $out/${tag}_rccl_20_runs.txt
    if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); cp $out/${tag}_rccl_run.log $out/${tag}_rccl_run_FAILED_$k.log; fi
  done
  echo "ok=$ok bad=$bad" | tee -a $out/${tag}_rccl_20_runs.txt
  grep "NCCL INFO" $out/${tag}_rccl_run.log | head -40 >> $out/${tag}_rccl_20_runs.txt
  ;;
*) echo "unknown step $step"; exit 2;;
esac
