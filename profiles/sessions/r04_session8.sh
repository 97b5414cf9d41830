#!/bin/bash
# round-4 GPU session 8: reduce kernels reworked; determinism tests; A/B incl. half the point splits
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "reproducible or backward or two_rank or fused_trainer" > gpurun_out/r4t8.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t8.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py r03:sanitize_grads=False default W2_BLOCKS-256 --rounds=3 > gpurun_out/r04_ab_slab_lambert.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_slab_lambert.txt | cut -c1-150
