#!/bin/bash
# round-4 GPU session 4: forward trunk schedule variants in one process: round 3 | anti-phase (default) | half lag, two passes | half lag, one pass (+ store policies, + GEMM priority)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 900 python profiles/ab_kernels.py r03:sanitize_grads=False default BN_GEMM_PRIO-1 BN_PP_HALF_LAG BN_PP_HALF_LAG_BN_PP_FUSED BN_PP_HALF_LAG_BN_PP_FUSED_BN_STASH_AUX-0 BN_PP_HALF_LAG_BN_PP_FUSED_BN_STASH_AUX-16 BN_PP_HALF_LAG_BN_PP_FUSED_BN_STASH_AUX-18 BN_PP_HALF_LAG_BN_PP_FUSED_BN_GEMM_PRIO-1 --rounds=3 > gpurun_out/r04_ab_fwd_variants.txt 2>&1; echo "ab rc=$?"
grep -n "field_fwd\|step (wall\|kernel " gpurun_out/r04_ab_fwd_variants.txt | cut -c1-400
