#!/bin/bash
# round-3 GPU session 2: launch-lean step - new unit tests, then the touched suites, then bench lines (lambert bf16 at 4096 and 512 rays)
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors_s2.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -m gpu -q > gpurun_out/r3t2_lean.log 2>&1
echo "lean rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t2_lean.log | cut -c1-300 | head -60
