#!/bin/bash
# round-4 GPU session 14: barrier-free backward trunk: weight-fragment prefetch depth 4 / 6, where the derivative loads are issued
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 700 python profiles/ab_kernels.py default BN_BWD_DEPTH-4 BN_BWD_DEPTH-6 BN_BWD_D_AT-1 BN_BWD_D_AT-2 BN_BWD_D_AT-2_BN_BWD_DEPTH-4 --rounds=3 > gpurun_out/r04_ab_bwd_pingpong_tune.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_bwd_pingpong_tune.txt | cut -c1-250
