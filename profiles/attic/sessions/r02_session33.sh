#!/bin/bash
# round-2 GPU session 33: half-mode fuzz
export BN_DIAG=$PWD/gpurun_out/r02_fuzz_errors.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k half > gpurun_out/t33.log 2>&1; rc=$?
tail -12 gpurun_out/t33.log | cut -c1-420
grep "fuzz-half" $BN_DIAG | sed 's/.*: rgb err/rgb err/' | sort -t' ' -k11 | head -5
exit $rc
