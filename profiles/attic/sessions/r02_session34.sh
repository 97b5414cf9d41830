#!/bin/bash
# round-2 GPU session 34: render-level fuzz
export BN_DIAG=$PWD/gpurun_out/r02_fuzz_errors.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k render > gpurun_out/t34.log 2>&1; rc=$?
tail -14 gpurun_out/t34.log | cut -c1-500
exit $rc
