#!/bin/bash
# round-4 GPU session 18: config 3 (RPV + analytic normals) per-kernel times, session 9's library against the barrier-free backward
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 600 python profiles/ab_kernels.py r04s9 default --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_bwd_pingpong_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -18 gpurun_out/r04_ab_bwd_pingpong_rpv_nan.txt | cut -c1-200
