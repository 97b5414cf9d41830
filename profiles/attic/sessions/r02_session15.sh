#!/bin/bash
# round-2 GPU session 15: sigma / learned-normal heads on the matrix pipe - tests, then A/B against the VALU-dot version
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors_g.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not psnr" > gpurun_out/t15.log 2>&1
tail -4 gpurun_out/t15.log
timeout -k 10 400 python profiles/ab_kernels.py prev_valu_sigma default --rounds=6 > gpurun_out/ab_sigma_mfma.txt 2>&1 || tail -5 gpurun_out/ab_sigma_mfma.txt
cat gpurun_out/ab_sigma_mfma.txt
timeout -k 10 400 python profiles/ab_kernels.py prev_valu_sigma default --rounds=4 --config=rpv_nlr > gpurun_out/ab_sigma_mfma_nlr.txt 2>&1 || tail -5 gpurun_out/ab_sigma_mfma_nlr.txt
cat gpurun_out/ab_sigma_mfma_nlr.txt
