#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
SVX_SKIP_SLOW=1 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -30 | tee gpurun_out/c6_pytest.txt
echo "== reader scaling"; python tools/reader_scaling.py 60000 2>&1 | grep -v "bamio\|amdgpu.ids" | tee gpurun_out/c6_reader_scaling.txt
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
echo "== bench c1"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c6_bench.json 2> gpurun_out/c6_bench.err
python - <<'PY'
import json
for l in open("gpurun_out/c6_bench.json"):
    if l.startswith("{"):
        j = json.loads(l); k = j["kernels"]
        print("ms/step %.2f" % j["ms_per_step"], "collect %.2f part %.2f edit %.2f link %.2f" % (k["collect_total_ms"], k["cluster_partition_sample_ms"], k["cluster_edit_distance_ms"], k["cluster_linkage_ms"]))
        print(json.dumps(j.get("end_to_end"), indent=1))
PY
echo "== c2 trace"
cd /tmp && rm -rf /tmp/kt && (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --workload c2 > /dev/null 2> /tmp/kt.err)
db=$(find /tmp/kt -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $db > gpurun_out/c6_timeline_c2.txt 2> gpurun_out/c6_timeline_c2.err
grep -n "k_" gpurun_out/c6_timeline_c2.txt | tail -52 | cut -c1-100
