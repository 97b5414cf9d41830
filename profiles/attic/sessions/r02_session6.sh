#!/bin/bash
# round-2 GPU session 6: variance of the RPV + analytic-normal PSNR gate under longer schedules
export BN_DIAG=$PWD/gpurun_out/r02_psnr_probe.txt
rm -f $BN_DIAG
BN_PSNR_PRE_STEPS=800 BN_PSNR_BRDF_STEPS=800 timeout -k 10 1000 python -m pytest tests -m gpu -q -k "psnr and rpv_nan" > gpurun_out/t6.log 2>&1
tail -3 gpurun_out/t6.log
cat $BN_DIAG
