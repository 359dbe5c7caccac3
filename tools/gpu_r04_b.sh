#!/bin/bash
# Round 4, second GPU pass: the device reader fault located (synchronous launches), shift-justified upper bounds A/B, 12-word band kernel, timeline.
tag=r04b
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "test_device_bam_decode_equals_host_reader and None" > gpurun_out/${tag}_pytest_reader_debug.txt 2>&1
grep -E "failed at|Error|passed|failed" gpurun_out/${tag}_pytest_reader_debug.txt | head -8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "edit_distance or cluster_scheduling or linkage_golden" > gpurun_out/${tag}_pytest_edit.txt 2>&1
tail -3 gpurun_out/${tag}_pytest_edit.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
for cfg in "0 0" "0 1" "1 0" "1 1"; do set -- $cfg
  SVX_EDIT_SHIFT_BOUNDS=$1 SVX_EDIT_NARROW=$2 timeout 300 python bench.py $B > gpurun_out/${tag}_bench_sb$1_nw$2.json 2> gpurun_out/${tag}_bench_sb$1_nw$2.err
done
python - <<'P'
import json
for sb in (0,1):
  for nw in (0,1):
    try:
        d=json.load(open('gpurun_out/r04b_bench_sb%d_nw%d.json'%(sb,nw)))
        k=d["kernels"]; print("shift_bounds",sb,"narrow",nw, "ms/step %.2f edit %.2f wc_issued %.3g wc_band %.3g frac_issue %.3f"%(d["ms_per_step"], k["cluster_edit_distance_ms"], k["edit_wordcols_issued"], k["edit_wordcols_band_kernels"], d["roofline_edit"]["frac_issue_cycles"]))
    except Exception as e: print(sb, nw, "failed", e)
P
SVX_EDIT_SERIAL=1 SVX_EDIT_PROFILE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> gpurun_out/${tag}_edit_profile_raw.txt
grep -E "edit_profile|edit_launch|edit_guess|edit_band_fit" gpurun_out/${tag}_edit_profile_raw.txt > gpurun_out/${tag}_edit_class_profile.jsonl; rm -f gpurun_out/${tag}_edit_profile_raw.txt
cd /tmp; rm -rf /tmp/kt && (cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null && python $R/tools/rocpd_timeline.py $db > $R/gpurun_out/${tag}_step_timeline.txt
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 --deselect "tests/test_gpu_parity.py::test_device_bam_decode_equals_host_reader" > gpurun_out/${tag}_pytest.txt 2>&1
tail -15 gpurun_out/${tag}_pytest.txt
