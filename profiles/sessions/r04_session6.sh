#!/bin/bash
# round-4 GPU session 6: full GPU suite on the slab-based (atomic-free, bitwise reproducible) weight gradients + new forward; A/B against round 3
export BN_DIAG=$PWD/gpurun_out/r04_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r4t6.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t6.log | grep -v "where\|+  " | cut -c1-250 | head -30
unset BN_DIAG
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py r03:sanitize_grads=False default --rounds=3 > gpurun_out/r04_ab_slab_lambert.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_slab_lambert.txt | cut -c1-120
