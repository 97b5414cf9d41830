#!/bin/bash
# round-4 GPU session 57: the affine chain GEMM in the backward chain alone: default (forward + backward) against bwd_noaff
# (-DBN_BWD_PP_AFFINE=0: forward only), alternating in one process
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py bwd_noaff default --config=lambert --rounds=5 > gpurun_out/r04_ab_bwd_affine_lambert.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_bwd_affine_lambert.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_full\|bwd_chain"
timeout -k 10 300 python profiles/ab_kernels.py bwd_noaff default --config=rpv_nan --rounds=4 > gpurun_out/r04_ab_bwd_affine_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -16 gpurun_out/r04_ab_bwd_affine_rpv_nan.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_full\|bwd_chain"
