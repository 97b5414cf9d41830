#!/bin/bash
# round-2 GPU session 49: accumulators initialised with the bias (16-bit forward) - correctness of the variant, then in-process A/B
BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_AB_BIAS_INIT/libbrdfnerf_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "half_modes" > gpurun_out/t49.log 2>&1; echo "variant half fuzz rc=$?"; tail -2 gpurun_out/t49.log
for cfg in lambert rpv_nan; do
  timeout -k 10 300 python profiles/ab_kernels.py default BN_AB_BIAS_INIT --config=$cfg --dtype=bf16 --rounds=7 > gpurun_out/ab49_$cfg.txt 2>&1 || { tail -5 gpurun_out/ab49_$cfg.txt; exit 1; }
  tail -9 gpurun_out/ab49_$cfg.txt
done
