#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
SVX_SKIP_SLOW=1 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 | tee gpurun_out/c2_pytest.txt
B="--steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end"
python bench.py $B > gpurun_out/c2_c1.json 2> gpurun_out/c2_c1.err
python bench.py $B --workload c2 > gpurun_out/c2_c2.json 2> gpurun_out/c2_c2.err
for pmd in 1000 5000 20000 100000; do
  python bench.py $B --workload c4 --partition-max-distance $pmd > gpurun_out/c2_c4_$pmd.json 2> gpurun_out/c2_c4_$pmd.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c2_c*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j["kernels"]; r = j.get("roofline_edit") or {}
        print(f, "ms/step %.2f first %.1f" % (j["ms_per_step"], j["first_step_ms"]), "reads/s %.3g" % j["value"], "collect %.2f part %.2f edit %.2f link %.2f scan %.3f" % (
            k["collect_total_ms"], k["cluster_partition_sample_ms"], k["cluster_edit_distance_ms"], k["cluster_linkage_ms"], k["k_cigar_scan_ms"]),
            "wc %.3g retry %.3g guess %s frac_cyc %.3f" % (k["edit_wordcols_issued"] or 0, k["edit_wordcols_retry_rounds"] or 0, k["edit_guess"], r.get("frac_issue_cycles", 0)),
            j["counts"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
cd /tmp && rm -rf /tmp/kt && (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $db > gpurun_out/c2_kernel_stats.csv 2>/dev/null
python tools/rocpd_timeline.py $db > gpurun_out/c2_timeline.txt 2>/dev/null
head -30 gpurun_out/c2_kernel_stats.csv
