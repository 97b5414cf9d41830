#!/bin/bash
# round-4 GPU session 7: buffer-store hazard fix (run-to-run identical stash), fuzz + determinism tests, A/B of the slab weight gradients
python profiles/dbg25b.py new 2>&1 | grep "stash bytes"
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "half_modes or fused_step or lean_step" > gpurun_out/r4t7.log 2>&1; echo "fuzz rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t7.log | cut -c1-250 | head
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "reproducible or backward or two_rank or fused_trainer" > gpurun_out/r4t7b.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t7b.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py r03:sanitize_grads=False default --rounds=3 > gpurun_out/r04_ab_slab_lambert.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_slab_lambert.txt | cut -c1-120
