#!/bin/bash
# round-2 GPU session 32: the full GPU suite again (fuzz first, as pytest orders the files), twice the PSNR part
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t32.log 2>&1
tail -5 gpurun_out/t32.log
grep -h "rpv_nan" $BN_DIAG | grep -i psnr | sed 's/.*paired/paired/; s/held-out PSNR rpv_nan, 600/600/' | cut -c1-330
