#!/bin/bash
# round-2 GPU session 30: differential fuzz of the field kernels over random configurations
export BN_DIAG=$PWD/gpurun_out/r02_fuzz_errors.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/t30.log 2>&1; rc=$?
tail -30 gpurun_out/t30.log
exit $rc
