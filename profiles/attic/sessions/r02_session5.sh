#!/bin/bash
# round-2 GPU session 5: PSNR gate (longer BRDF stage), new pin tests, whole suite
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/t5.log 2>&1
tail -15 gpurun_out/t5.log
grep -i "psnr\|trajectory" $BN_DIAG
