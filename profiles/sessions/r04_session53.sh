#!/bin/bash
# round-4 GPU session 53: chain GEMM with one LDS base address per point tile and block of k-steps (-DBN_GEMM_AFFINE_B: 4 instead
# of 24 vector adds per 48 MFMAs; more registers in some kernels) against the default
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py default affine --config=lambert --rounds=4 > gpurun_out/r04_ab_affine_lambert.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_affine_lambert.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_\|bwd_chain"
timeout -k 10 300 python profiles/ab_kernels.py default affine --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_affine_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -16 gpurun_out/r04_ab_affine_rpv_nan.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_\|bwd_chain\|adjoint"
