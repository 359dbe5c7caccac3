#!/bin/bash
# Round 4, eleventh GPU pass: multi-window inflate steps (up to 4 windows) against 2 and 3; the reader tests on the device; a short bench with the end-to-end block.
tag=r04k
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_inflate_rate.txt; grep "GPU\|identical" gpurun_out/${tag}_bgzf_inflate_rate.txt
for v in win2 win3; do
  SVX_LIB=svim_amd/variants/libsvx_$v.so python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_inflate_rate_$v.txt; echo "-- $v"; grep "GPU" gpurun_out/${tag}_bgzf_inflate_rate_$v.txt
done
SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py 40000 qual 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_inflate_profile_qual.txt; cat gpurun_out/${tag}_inflate_profile_qual.txt
SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_inflate_profile.txt; head -3 gpurun_out/${tag}_inflate_profile.txt; tail -2 gpurun_out/${tag}_inflate_profile.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_foreign_bam.py -m gpu -q -x -p no:cacheprovider -k "bam or inflate or bgzf or reader" > gpurun_out/${tag}_pytest_reader.txt 2>&1
tail -3 gpurun_out/${tag}_pytest_reader.txt
( time timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err ) 2>&1 | grep real
python - <<'P'
import json
try:
    d=json.load(open('gpurun_out/r04k_bench_c1.json'))
    k=d["kernels"]; print("ms/step %.2f edit %.2f first %.1f value %.3g"%(d["ms_per_step"], k["cluster_edit_distance_ms"], d["first_step_ms"], d["value"]))
    e=d.get("end_to_end",{}); print({x:e.get(x) for x in ("bam_file_reads_per_s","bam_file_first_pass_reads_per_s","objects_materialised_reads_per_s","bam_file_host_decode_reads_per_s","error")})
    print((e.get("bam_file") or {}).get("inflate_kernel_MB_per_s"), (e.get("bam_file_without_base_qualities") or {}).get("reads_per_s"), (e.get("bam_file_without_base_qualities") or {}).get("inflate_kernel_MB_per_s"))
except Exception as ex: print("bench failed", ex); print(open('gpurun_out/r04k_bench_c1.err').read()[-1500:])
P
