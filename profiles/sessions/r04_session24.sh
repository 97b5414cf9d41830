#!/bin/bash
# round-4 GPU session 24: backward chain with analytic normals: all zbar pieces of a layer fetched together at the top of the epilogue
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or normal" > gpurun_out/r4t24.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t24.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py r04s22 default --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_zbar_prefetch_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -18 gpurun_out/r04_ab_zbar_prefetch_rpv_nan.txt | cut -c1-120
timeout -k 10 300 python profiles/ab_kernels.py r04s22 default --rounds=2 > gpurun_out/r04_ab_zbar_prefetch_lambert.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_zbar_prefetch_lambert.txt | cut -c1-120
