#!/bin/bash
# round-2 GPU session 46: one-off fuzz hunt with 5 x the committed seeds
export BN_FUZZ_SCALE=15 BN_DIAG=$PWD/gpurun_out/fuzz_hunt.txt
rm -f $BN_DIAG
timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/t46.log 2>&1; rc=$?
tail -25 gpurun_out/t46.log | cut -c1-420
exit 0
