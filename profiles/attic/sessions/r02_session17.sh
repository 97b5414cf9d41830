#!/bin/bash
# round-2 GPU session 17: PSNR gates on a geometrically consistent scene
export BN_DIAG=$PWD/gpurun_out/r02_psnr_probe.txt
rm -f $BN_DIAG
timeout -k 10 1100 python -m pytest tests -m gpu -q -k "psnr" > gpurun_out/t17.log 2>&1
tail -3 gpurun_out/t17.log
cat $BN_DIAG
