#!/bin/bash
# round-3 GPU session 1: D8 derivative stash + per-instantiation prefetch depth + -fno-slp-vectorize: GPU suite, then A/B against
# the round-2 library (profiles/build_baseline.py HEAD r02base) in one process
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors_s1.txt
rm -f $BN_DIAG
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r3t1.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r3t1.log | cut -c1-250
for cfg in lambert rpv_nan; do
  timeout -k 10 240 python profiles/ab_kernels.py r02base default --config=$cfg --dtype=bf16 --rounds=5 > gpurun_out/r3ab1_${cfg}_bf16.txt 2>&1 || { echo "ab $cfg failed"; tail -5 gpurun_out/r3ab1_${cfg}_bf16.txt; }
  tail -25 gpurun_out/r3ab1_${cfg}_bf16.txt
done
timeout -k 10 240 python profiles/ab_kernels.py r02base default --config=lambert --dtype=fp16 --rounds=5 > gpurun_out/r3ab1_lambert_fp16.txt 2>&1 || { echo "ab fp16 failed"; tail -5 gpurun_out/r3ab1_lambert_fp16.txt; }
tail -25 gpurun_out/r3ab1_lambert_fp16.txt
