#!/bin/bash
# round-4 GPU session 27: --noise_std on the launch-lean step (ABI 6: bn_noise, bn_rng_normal)
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -q -m gpu -x > gpurun_out/r4t27.log 2>&1; echo "lean rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t27.log | cut -c1-250 | head -20
