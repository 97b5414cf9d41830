#!/bin/bash
# round-4 GPU session 28: full GPU suite with --noise_std on the lean step (ABI 6) and the fuzz drawing it
export BN_DIAG=$PWD/gpurun_out/r04_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4t28.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t28.log | grep -v "where\|+  " | cut -c1-250 | head -30
grep -c "noise=0.4" $BN_DIAG
