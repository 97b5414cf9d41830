#!/bin/bash
# round-4 GPU session 2: two-set B fragments in the chain GEMM x anti-phase trunk: parity subset, A/B of the four combinations + round 3, timeline
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "field_forward or half or folding or fused_trainer_matches or device_fault or field_backward" > gpurun_out/r4t2.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t2.log | cut -c1-300 | head -20
BRDFNERF_ALLOW_STALE_LIB=1 timeout -k 10 600 python profiles/ab_kernels.py r03:sanitize_grads=False default BN_GEMM_B1 BN_PP_HALF_LAG BN_PP_HALF_LAG_BN_GEMM_B1 --rounds=3 > gpurun_out/r04_ab_antiphase_lambert.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_antiphase_lambert.txt
timeout -k 10 200 python profiles/simd_timeline.py --no-build > gpurun_out/r04_simd_timeline.txt 2>&1; echo "tl rc=$?"
tail -3 gpurun_out/r04_simd_timeline.txt
