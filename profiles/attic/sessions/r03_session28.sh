#!/bin/bash
# round-3 GPU session 28: NormalRegLoss inside the merged-set compositing kernels (lean step) - kernel test, lean vs general, lean fuzz
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -q -m gpu > gpurun_out/r3t28.log 2>&1; echo "lean rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t28.log | cut -c1-300 | head -20
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "lean_step or fused_step" > gpurun_out/r3t28b.log 2>&1; echo "fuzz rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t28b.log | cut -c1-300 | head -20
