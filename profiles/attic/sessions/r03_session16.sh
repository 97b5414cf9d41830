#!/bin/bash
# round-3 GPU session 16: launch-lean tests incl. bn_ray_shade_loss on the reference's per-sample outputs; TrainLoop with staging buffers
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -q -m gpu > gpurun_out/r3t16.log 2>&1; echo "lean rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t16.log | cut -c1-250 | head -20
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "train_loop or TrainLoop or ray_table or resume or checkpoint" > gpurun_out/r3t16b.log 2>&1; echo "loop rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t16b.log | cut -c1-250 | head -20
