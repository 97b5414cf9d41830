#!/bin/bash
# round-2 GPU session 36: fused-step fuzz
export BN_DIAG=$PWD/gpurun_out/r02_fuzz_errors.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "fused_step_half" > gpurun_out/t36.log 2>&1; rc=$?
tail -16 gpurun_out/t36.log | cut -c1-600
exit $rc
