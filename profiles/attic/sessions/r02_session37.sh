#!/bin/bash
# round-2 GPU session 37: the whole fuzz file with F = 512 among the random widths
export BN_DIAG=$PWD/gpurun_out/r02_fuzz_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/t37.log 2>&1; rc=$?
tail -14 gpurun_out/t37.log | cut -c1-500
exit $rc
