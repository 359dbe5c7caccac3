#!/bin/bash
# Round 4, sixth GPU pass: the torch / libsvx stream race in the multi-GPU fetches fixed? (8 ranks repeated), two-window inflate steps A/B, bench, whole suite.
tag=r04f
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
run_c3() {
  label=$1; shift
  port=$((29600 + RANDOM % 300))
  out=$(env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port tests/mp_c3_ranks_one_gpu.py /tmp/wg_$label.bam 0.003 3 2>/dev/null | grep -E "^C3_RANKS")
  echo "== $label: $out" | tee -a gpurun_out/${tag}_c3_repeat.txt
}
for i in 1 2 3; do run_c3 default_$i X=1; done
python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_inflate_rate.txt; grep "GPU" gpurun_out/${tag}_bgzf_inflate_rate.txt
if [ -f svim_amd/variants/libsvx_nowide.so ]; then
  SVX_LIB=svim_amd/variants/libsvx_nowide.so python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_inflate_rate_one_window.txt; grep "GPU" gpurun_out/${tag}_bgzf_inflate_rate_one_window.txt
fi
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err ) 2>&1 | grep real
python - <<'P'
import json
try:
    d=json.load(open('gpurun_out/r04f_bench_c1.json'))
    k=d["kernels"]; print("ms/step %.2f edit %.2f first %.1f value %.3g"%(d["ms_per_step"], k["cluster_edit_distance_ms"], d["first_step_ms"], d["value"]))
    e=d.get("end_to_end",{}); print({x:e.get(x) for x in ("bam_file_reads_per_s","bam_file_first_pass_reads_per_s","objects_materialised_reads_per_s","bam_file_host_decode_reads_per_s","error")})
    print((e.get("bam_file") or {}).get("inflate_kernel_MB_per_s"), (e.get("bam_file_without_base_qualities") or {}).get("reads_per_s"), (e.get("bam_file_without_base_qualities") or {}).get("inflate_kernel_MB_per_s"))
except Exception as ex: print("bench failed", ex); print(open('gpurun_out/r04f_bench_c1.err').read()[-1500:])
P
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/${tag}_pytest.txt 2>&1
tail -12 gpurun_out/${tag}_pytest.txt
