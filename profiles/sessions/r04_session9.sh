#!/bin/bash
# round-4 GPU session 9: full GPU suite (gsam_only lean step, lean step vs oracle at F = 512, end-to-end cosines reported)
export BN_DIAG=$PWD/gpurun_out/r04_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r4t9.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t9.log | grep -v "where\|+  " | cut -c1-250 | head -30
grep -n "END-TO-END\|lean step vs oracle\|gsam_only step" $BN_DIAG | cut -c1-400
