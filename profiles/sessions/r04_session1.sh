#!/bin/bash
# round-4 GPU session 1: anti-phase forward trunk - parity of the 16-bit forward paths, A/B against the round-3 library, SIMD timelines
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "field_forward or half or folding or fused_trainer_matches or device_fault" > gpurun_out/r4t1.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t1.log | cut -c1-300 | head -20
BRDFNERF_ALLOW_STALE_LIB=1 timeout -k 10 300 python profiles/ab_kernels.py r03 default --rounds=3 > gpurun_out/r04_ab_antiphase_lambert.txt 2>&1; echo "ab rc=$?"
tail -12 gpurun_out/r04_ab_antiphase_lambert.txt
timeout -k 10 200 python profiles/simd_timeline.py --no-build > gpurun_out/r04_simd_timeline.txt 2>&1; echo "tl rc=$?"
tail -3 gpurun_out/r04_simd_timeline.txt
timeout -k 10 200 python profiles/simd_timeline.py -DBN_PP_HALF_LAG --no-build > gpurun_out/r04_simd_timeline_halflag.txt 2>&1; echo "tl2 rc=$?"
tail -3 gpurun_out/r04_simd_timeline_halflag.txt
