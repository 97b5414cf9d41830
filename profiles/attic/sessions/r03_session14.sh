#!/bin/bash
# round-3 GPU session 14: the full-size tests with the seeded 16-bit field backward, then validation part b
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors_fullsize.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_size or train_loop or two_rank" > gpurun_out/r3t14.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t14.log | grep -v "where\|+  " | cut -c1-250 | head -30
grep -h "started from" $BN_DIAG | cut -c1-250
unset BN_DIAG
bash profiles/sessions/r03_session13b.sh
