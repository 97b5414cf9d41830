#!/bin/bash
# round-2 GPU session 47: the two adjusted fuzz tests at 15 x seeds, then the file at its committed size
export BN_DIAG=$PWD/gpurun_out/fuzz_hunt.txt
rm -f $BN_DIAG
BN_FUZZ_SCALE=15 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "brdf_inputs or configuration_against_oracle" > gpurun_out/t47a.log 2>&1
tail -6 gpurun_out/t47a.log | cut -c1-400
grep "^E  *AssertionError" gpurun_out/t47a.log | cut -c1-400
timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/t47.log 2>&1
tail -3 gpurun_out/t47.log
