#!/bin/bash
# round-4 GPU session 5: new default forward (half lag, one pass, bias-started accumulators, buffer stores, GEMM priority 1): parity subset, priority levels, timeline
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "field_forward or half or folding or fused_trainer_matches or device_fault or field_backward or lambert" > gpurun_out/r4t5.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t5.log | cut -c1-300 | head -20
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 900 python profiles/ab_kernels.py r03:sanitize_grads=False default BN_GEMM_PRIO-0 BN_GEMM_PRIO-2 BN_GEMM_PRIO-3 BN_PP_FULL_LAG --rounds=3 > gpurun_out/r04_ab_fwd_prio.txt 2>&1; echo "ab rc=$?"
grep -n "field_fwd\|step (wall\|kernel " gpurun_out/r04_ab_fwd_prio.txt | cut -c1-400
unset BRDFNERF_ALLOW_STALE_LIB
timeout -k 10 200 python profiles/simd_timeline.py --no-build > gpurun_out/r04_simd_timeline.txt 2>&1; echo "tl rc=$?"
tail -3 gpurun_out/r04_simd_timeline.txt
