#!/bin/bash
# round-4 GPU session 13: backward chain with native-order dZ stores from the epilogue registers (weight gradient stages native A
# chunks) and a barrier-free trunk like the forward's: parity of the backward / trainer tests, then A/B against session 9's library
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or fused_trainer or reproducible or two_rank" > gpurun_out/r4t13.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t13.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py r04s9 default BN_BWD_NO_PINGPONG --rounds=3 > gpurun_out/r04_ab_bwd_pingpong.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_bwd_pingpong.txt | cut -c1-200
