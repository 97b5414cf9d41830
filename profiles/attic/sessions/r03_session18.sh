#!/bin/bash
# round-3 GPU session 18: non-finite rays left out of the step (bn_ray_shade_loss nonfinite) - lean tests, full-size tests, soak
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -q -m gpu > gpurun_out/r3t18.log 2>&1; echo "lean rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t18.log | cut -c1-250 | head -20
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_size or beta" > gpurun_out/r3t18b.log 2>&1; echo "full-size rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t18b.log | cut -c1-250 | head -20
timeout -k 10 300 python profiles/soak.py 1500 > gpurun_out/r03_soak.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03_soak.txt | cut -c1-400
