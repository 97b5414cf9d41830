#!/bin/bash
# round-2 GPU session 20: --beta parity (six-head layout), A/B of the backward with the four-head DPH stride, full GPU suite
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "beta or viewdir or fused_trainer_matches" > gpurun_out/t20a.log 2>&1; rc=$?
tail -25 gpurun_out/t20a.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python profiles/ab_kernels.py default BN_DPH-12 --config=lambert --rounds=5 > gpurun_out/ab20_lambert.txt 2>&1 || exit 1
tail -8 gpurun_out/ab20_lambert.txt
timeout -k 10 200 python profiles/ab_kernels.py default BN_DPH-12 --config=rpv_nan --rounds=5 > gpurun_out/ab20_rpv.txt 2>&1 || exit 1
tail -10 gpurun_out/ab20_rpv.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t20.log 2>&1
tail -8 gpurun_out/t20.log
