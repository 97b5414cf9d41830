#!/bin/bash
# round-3 GPU session 22: lean-step fuzz, TrainLoop graph replay, full-size tests
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "lean_step" > gpurun_out/r3t22.log 2>&1; echo "fuzz-lean rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t22.log | cut -c1-300 | head -20
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "train_loop or full_size or beta" > gpurun_out/r3t22b.log 2>&1; echo "loop rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t22b.log | cut -c1-300 | head -20
