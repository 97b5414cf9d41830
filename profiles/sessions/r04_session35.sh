#!/bin/bash
# round-4 GPU session 35: wgrad256 with 4 waves per workgroup (wave tile 128 x 128, accumulators in AGPRs, one wave per SIMD)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 env BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/W2_WAVES-4/libbrdfnerf_hip.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or reproducible" > gpurun_out/r4t35.log 2>&1; echo "parity (4 waves) rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t35.log | cut -c1-250 | head
timeout -k 10 400 python profiles/ab_kernels.py default W2_WAVES-4 --rounds=3 > gpurun_out/r04_ab_wgrad_4waves.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_wgrad_4waves.txt | cut -c1-100 | grep -v "pack\|composite\|guided\|strat\|adam\|skinny\|reduce"
