#!/bin/bash
# round-2 GPU session 18: --input_viewdir parity, A/B of the forward with the direction segment compiled out, full GPU suite
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "viewdir" > gpurun_out/t18a.log 2>&1; rc=$?
tail -15 gpurun_out/t18a.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python profiles/ab_kernels.py default BN_AB_NO_VIEWDIR --config=lambert --rounds=5 > gpurun_out/ab18_lambert.txt 2>&1 || exit 1
tail -12 gpurun_out/ab18_lambert.txt
timeout -k 10 200 python profiles/ab_kernels.py default BN_AB_NO_VIEWDIR --config=rpv_nan --rounds=5 > gpurun_out/ab18_rpv.txt 2>&1 || exit 1
tail -12 gpurun_out/ab18_rpv.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t18.log 2>&1
tail -8 gpurun_out/t18.log
