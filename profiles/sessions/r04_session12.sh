#!/bin/bash
# round-4 GPU session 12: backward chain: derivative pieces fetched a layer ahead (behind the previous GEMM's last weight
# fragment, barriers that wait for LDS only), weight-fragment prefetch depth 4 / 6
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 600 python profiles/ab_kernels.py default BN_BWD_D_AHEAD BN_BWD_DEPTH-4 BN_BWD_DEPTH-6 BN_BWD_D_AHEAD_BN_BWD_DEPTH-4 --rounds=3 > gpurun_out/r04_ab_bwd_dahead.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_bwd_dahead.txt | cut -c1-220
timeout -k 10 300 env BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_BWD_D_AHEAD/libbrdfnerf_hip.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer" > gpurun_out/r4t12.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t12.log | cut -c1-250 | head
