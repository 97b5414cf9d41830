#!/bin/bash
# round-4 GPU session 37: the whole differential fuzz at three times its default length (new seeds on the final kernels)
BN_FUZZ_SCALE=3 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -q -m gpu > gpurun_out/r4t37.log 2>&1; echo "fuzz x3 rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t37.log | cut -c1-300 | head -20
