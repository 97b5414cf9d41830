#!/bin/bash
# round-4 GPU session 26: full GPU suite (zbar / adjoint-backward load batching, pinned wgrad prologue order)
export BN_DIAG=$PWD/gpurun_out/r04_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4t26.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t26.log | grep -v "where\|+  " | cut -c1-250 | head -30
