#!/bin/bash
# round-2 GPU session 24: scheduler-strategy variants of the whole library, in-process A/B (bf16 and fp16, lambert and rpv_nan)
for cfg in lambert rpv_nan; do
  for dt in bf16 fp16; do
    timeout -k 10 300 python profiles/ab_kernels.py default maxilp nomisched --config=$cfg --dtype=$dt --rounds=5 > gpurun_out/ab24_${cfg}_$dt.txt 2>&1 || { tail -5 gpurun_out/ab24_${cfg}_$dt.txt; exit 1; }
    tail -9 gpurun_out/ab24_${cfg}_$dt.txt
  done
done
