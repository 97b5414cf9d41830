#!/bin/bash
# round-4 GPU session 25: adjoint backward: derivative bytes before the GEMM, a_{l+1} / y_l pieces in batches (2 / 4 point tiles)
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or normal" > gpurun_out/r4t25.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t25.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py r04s22 default ADJ_GRP-4 --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_adjbwd_batches_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -18 gpurun_out/r04_ab_adjbwd_batches_rpv_nan.txt | cut -c1-120
