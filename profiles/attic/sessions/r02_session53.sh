#!/bin/bash
# round-2 GPU session 53: field fuzz with the fp64 referee, 15 x seeds; then the committed file
BN_FUZZ_SCALE=15 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "configuration_against_oracle" > gpurun_out/t53a.log 2>&1
tail -3 gpurun_out/t53a.log | cut -c1-300
grep "^E  *AssertionError" gpurun_out/t53a.log | cut -c1-400
timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/t53.log 2>&1
tail -2 gpurun_out/t53.log
