#!/bin/bash
# round-2 GPU session 25: the split weight-gradient unit (max-ILP scheduler) - determinism + wgrad parity tests, then in-process
# A/B of three more scheduler options on the whole library
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "deterministic or field_backward or half or fused_trainer_matches or beta or viewdir" > gpurun_out/t25a.log 2>&1; rc=$?
tail -4 gpurun_out/t25a.log
[ $rc -eq 0 ] || exit $rc
for cfg in lambert rpv_nan; do
  for dt in bf16 fp16; do
    timeout -k 10 300 python profiles/ab_kernels.py default trackers nounclust noclust --config=$cfg --dtype=$dt --rounds=5 > gpurun_out/ab25_${cfg}_$dt.txt 2>&1 || { tail -5 gpurun_out/ab25_${cfg}_$dt.txt; exit 1; }
    tail -9 gpurun_out/ab25_${cfg}_$dt.txt
  done
done
