#!/bin/bash
# round-2 GPU session 42: re-tune two grid / pipeline constants on the per-unit scheduler build (in-process A/B)
for cfg in lambert rpv_nan; do
  timeout -k 10 400 python profiles/ab_kernels.py default W2_BLOCKS-384 W2_BLOCKS-768 BN_DEPTH_BF16-2 BN_DEPTH_BF16-6 --config=$cfg --dtype=bf16 --rounds=5 > gpurun_out/ab42_$cfg.txt 2>&1 || { tail -5 gpurun_out/ab42_$cfg.txt; exit 1; }
  tail -9 gpurun_out/ab42_$cfg.txt
done
