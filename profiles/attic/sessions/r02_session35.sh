#!/bin/bash
# round-2 GPU session 35: all fuzz tests, then the full GPU suite
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/t35.log 2>&1; rc=$?
tail -6 gpurun_out/t35.log | cut -c1-300
exit $rc
