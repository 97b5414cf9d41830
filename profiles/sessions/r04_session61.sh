#!/bin/bash
# round-4 GPU session 61 (measurement for the next round; NOT adopted - no budget left to re-validate): the half-GEMMs of the
# barrier-free trunks (16 k-steps at F = 512) as straight-line code (-DBN_PP_FIXED_NKS: no clamps, no tail steps behind
# run-time conditions - the tails copy the whole accumulator set with ~70 v_mov_b64 per half-GEMM in the product ISA)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 150 env BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/ppfixed/libbrdfnerf_hip.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "field_forward or field_backward or fused_trainer" > gpurun_out/r4t61.log 2>&1; echo "parity (ppfixed) rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t61.log | cut -c1-200 | head -5
timeout -k 10 150 python profiles/ab_kernels.py default ppfixed --config=lambert --rounds=4 > gpurun_out/r04_ab_pp_fixed_nks.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_pp_fixed_nks.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_\|bwd_chain"
