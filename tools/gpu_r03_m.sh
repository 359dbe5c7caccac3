#!/bin/bash
export TMPDIR=/tmp HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out
SVX_LIB=svim_amd/variants/libsvx_prof.so timeout 300 python tools/inflate_profile.py 60000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03m_inflate_profile.txt
SVX_INFLATE_LDS_PAD=24000 SVX_LIB=svim_amd/variants/libsvx_prof.so timeout 300 python tools/inflate_profile.py 60000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03m_inflate_profile_5waves.txt
