#!/bin/bash
# round-4 GPU session 30: adjoint chain with native delta stores and without workgroup barriers below the seed
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or normal or render" > gpurun_out/r4t30.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t30.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py r04s29 default --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_adjoint_pingpong_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -18 gpurun_out/r04_ab_adjoint_pingpong_rpv_nan.txt | cut -c1-120
