#!/bin/bash
# round-4 GPU session 20: barrier-free wgrad256 (ring of four 32-point slots, LDS counters): parity, then A/B against the barrier form
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or reproducible or two_rank" > gpurun_out/r4t20.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t20.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py W2_NO_RING default --rounds=3 > gpurun_out/r04_ab_wgrad_ring.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_wgrad_ring.txt | cut -c1-200
