#!/bin/bash
# Round 4, fourth GPU pass: query-name mode on the device reader, leaner hand-off in the multi-lane full-matrix kernels, the end_to_end block on the file with
# base qualities, inflate cycle profile on literal-heavy blocks, whole suite.
tag=r04d
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "queryname or edit_distance or bam_path_native" > gpurun_out/${tag}_pytest_first.txt 2>&1
tail -4 gpurun_out/${tag}_pytest_first.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err ) 2>&1 | grep real
python - <<'P'
import json
try:
    d=json.load(open('gpurun_out/r04d_bench_c1.json'))
    k=d["kernels"]; print("ms/step %.2f edit %.2f first %.1f value %.3g"%(d["ms_per_step"], k["cluster_edit_distance_ms"], d["first_step_ms"], d["value"]))
    e=d.get("end_to_end",{}); print({x:e.get(x) for x in ("bam_file_reads_per_s","bam_file_first_pass_reads_per_s","objects_materialised_reads_per_s","bam_file_host_decode_reads_per_s","error")})
    print((e.get("bam_file") or {}).get("inflate_kernel_MB_per_s"), (e.get("bam_file_without_base_qualities") or {}).get("reads_per_s"), e.get("sample"))
    c=d.get("cpu_baseline",{}); print(c.get("value"), c.get("reference_python_reads_per_s"), c.get("parity_vs_gpu_on_sample"))
except Exception as ex: print("bench failed", ex); print(open('gpurun_out/r04d_bench_c1.err').read()[-1500:])
P
python tools/bgzf_inflate_rate.py 60000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_bgzf_inflate_rate.txt; tail -6 gpurun_out/${tag}_bgzf_inflate_rate.txt
if [ -f svim_amd/variants/libsvx_prof.so ]; then
  SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py 40000 qual 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_inflate_profile_qual.txt
  SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py 40000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_inflate_profile.txt
  cat gpurun_out/${tag}_inflate_profile_qual.txt
fi
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > gpurun_out/${tag}_pytest.txt 2>&1
tail -16 gpurun_out/${tag}_pytest.txt
