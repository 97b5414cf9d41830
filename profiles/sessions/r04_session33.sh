#!/bin/bash
# round-4 GPU session 33: the skinny weight-gradient jobs on a second stream beside wgrad256 (a fork / join in the captured step)
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py tests/test_gpu_parity.py -q -m gpu -x -k "lean_step or fused_trainer or reproducible or two_rank or train_loop" > gpurun_out/r4t33.log 2>&1; echo "tests rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t33.log | cut -c1-250 | head
timeout -k 10 400 python profiles/ab_kernels.py default:side_skinny=False default:side_skinny=True --rounds=4 > gpurun_out/r04_ab_side_skinny_lambert.txt 2>&1; echo "ab rc=$?"
tail -3 gpurun_out/r04_ab_side_skinny_lambert.txt | cut -c1-120
timeout -k 10 400 python profiles/ab_kernels.py default:side_skinny=False default:side_skinny=True --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_side_skinny_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -3 gpurun_out/r04_ab_side_skinny_rpv_nan.txt | cut -c1-120
timeout -k 10 300 python profiles/ab_kernels.py default:side_skinny=False default:side_skinny=True --rays=512 --rounds=4 > gpurun_out/r04_ab_side_skinny_512.txt 2>&1; echo "ab rc=$?"
tail -3 gpurun_out/r04_ab_side_skinny_512.txt | cut -c1-120
