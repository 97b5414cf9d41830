#!/bin/bash
# round-3 GPU session 25: the full GPU suite + smoke on the final tree
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3t25.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t25.log | grep -v "where\|+  " | cut -c1-250 | head -30
unset BN_DIAG
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r3smoke25.log 2>&1; tail -2 gpurun_out/r3smoke25.log
