#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
SVX_SKIP_SLOW=1 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -30 | tee gpurun_out/c3_pytest.txt
B="--steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end"
python bench.py $B > gpurun_out/c3_c1.json 2> gpurun_out/c3_c1.err
SVX_BENCH_FORCE_DIST=1 python bench.py $B > gpurun_out/c3_c1_dist.json 2> gpurun_out/c3_c1_dist.err
python bench.py $B --workload c2 > gpurun_out/c3_c2.json 2> gpurun_out/c3_c2.err
SVX_BENCH_FORCE_DIST=1 python bench.py $B --workload c2 > gpurun_out/c3_c2_dist.json 2> gpurun_out/c3_c2_dist.err
python bench.py $B --workload c4 > gpurun_out/c3_c4.json 2> gpurun_out/c3_c4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c3_c*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j["kernels"]; r = j.get("roofline_edit") or {}
        print(f, "ms/step %.2f first %.1f" % (j["ms_per_step"], j["first_step_ms"]), "reads/s %.3g" % j["value"], "collect %.2f part %.2f edit %.2f link %.2f scan %.3f" % (
            k["collect_total_ms"], k["cluster_partition_sample_ms"], k["cluster_edit_distance_ms"], k["cluster_linkage_ms"], k["k_cigar_scan_ms"]),
            "wc %.3g retry %.3g guess %s frac_cyc %.3f" % (k["edit_wordcols_issued"] or 0, k["edit_wordcols_retry_rounds"] or 0, k["edit_guess"], r.get("frac_issue_cycles", 0)))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-1500:])
PY
cd /tmp && rm -rf /tmp/kt && (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --workload c2 > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $db > gpurun_out/c3_timeline_c2.txt 2>/dev/null
grep -n "k_" gpurun_out/c3_timeline_c2.txt | tail -50 | cut -c1-100
