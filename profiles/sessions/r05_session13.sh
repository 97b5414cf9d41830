#!/bin/bash
# round-5 GPU session 13: the clock the chip holds inside the three fused kernels of the final sources (s_memtime / s_memrealtime stamps,
# diagnostic builds), to read the timeline's cycle counts in time
O=gpurun_out
export BRDFNERF_ALLOW_STALE_LIB=1
BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_CLOCK_STAMP/libbrdfnerf_hip.so timeout -k 10 200 python profiles/clock_probe.py > $O/r05_clock_probe.txt 2>&1; echo "rc=$?"
BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_CLOCK_STAMP_BN_CLOCK_STAMP_WGRAD/libbrdfnerf_hip.so timeout -k 10 200 python profiles/clock_probe.py >> $O/r05_clock_probe.txt 2>&1; echo "rc=$?"
cat $O/r05_clock_probe.txt | grep -v amdgpu.ids | cut -c1-200
