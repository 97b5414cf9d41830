#!/bin/bash
# round-4 GPU session 21: wgrad256 with the prologue's prefetch order pinned (counted vmcnt waits in the stage loop instead of
# vmcnt(0)), native chunks staged with two ds_write_b64 / with the swap + ds_write_b128: parity, then A/B against session 13's library
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or reproducible or two_rank" > gpurun_out/r4t21.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t21.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 env BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/W2_SWAP_B128/libbrdfnerf_hip.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer" > gpurun_out/r4t21b.log 2>&1; echo "parity (swap) rc=$?"
timeout -k 10 500 python profiles/ab_kernels.py r04s9 W2_SWAP_B128 default --rounds=3 > gpurun_out/r04_ab_wgrad_waits.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_wgrad_waits.txt | cut -c1-200
