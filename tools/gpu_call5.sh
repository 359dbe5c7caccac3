#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== read bandwidth ceiling"; tools/micro/read_bw.bin 5.7 2>&1 | tee gpurun_out/c5_read_bw.txt
echo "== scan sweep"
for v in tree nu3 nu4; do
  lib=svim_amd/variants/libsvx_$v.so; [ "$v" = "tree" ] && lib=svim_amd/libsvx.so
  echo "variant $v"; SVX_LIB=$PWD/$lib python tools/scan_sweep.py 2>&1 | grep "^map"
done | tee gpurun_out/c5_scan_sweep.txt
echo "== bench c1"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c5_bench.json 2> gpurun_out/c5_bench.err
python - <<'PY'
import json
for l in open("gpurun_out/c5_bench.json"):
    if l.startswith("{"):
        j = json.loads(l); k = j["kernels"]
        print("ms/step %.2f" % j["ms_per_step"], "collect %.2f part %.2f edit %.2f link %.2f" % (k["collect_total_ms"], k["cluster_partition_sample_ms"], k["cluster_edit_distance_ms"], k["cluster_linkage_ms"]))
        e = j.get("end_to_end") or {}
        print("e2e bam %s dense %s host_arrays %s" % (e.get("bam_file_reads_per_s"), e.get("bam_file_dense_seq_reads_per_s"), e.get("host_arrays_reads_per_s")), e.get("bam_file"), e.get("error"))
PY
echo "== reader scaling"; python tools/reader_scaling.py 60000 2>&1 | grep -v bamio | tee gpurun_out/c5_reader_scaling.txt
echo "== c2 trace"
cd /tmp && rm -rf /tmp/kt && (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --workload c2 > /dev/null 2> /tmp/kt.err)
find /tmp/kt -type f | head; tail -3 /tmp/kt.err
db=$(find /tmp/kt -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $db > gpurun_out/c5_timeline_c2.txt 2> gpurun_out/c5_timeline_c2.err; tail -3 gpurun_out/c5_timeline_c2.err
grep -n "k_" gpurun_out/c5_timeline_c2.txt | tail -52 | cut -c1-100
