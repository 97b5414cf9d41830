#!/bin/bash
# round-5 GPU session 10: the per-layer scalars of the argument struct (packed offsets, bias pointer, stash arrays) fetched one layer
# ahead in the two trunk kernels: quick parity of the chain kernels, then A/B against the product form (lambert and rpv_nan)
O=gpurun_out
export BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_LAYER_AHEAD/libbrdfnerf_hip.so BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py -m gpu -x -q -k "forward or backward or lean_step_matches or reproducible" > $O/r05_s10_pytest.log 2>&1; rc=$?; echo "pytest (variant library) rc=$rc"; tail -3 $O/r05_s10_pytest.log | cut -c1-200
unset BRDFNERF_HIP_LIB
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python profiles/ab_kernels.py BN_LAYER_AHEAD default --config=lambert --rounds=4 > $O/r05_ab_layer_ahead_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_layer_ahead_lambert.txt | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python profiles/ab_kernels.py BN_LAYER_AHEAD default --config=rpv_nan --rounds=3 > $O/r05_ab_layer_ahead_rpv_nan.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -20 $O/r05_ab_layer_ahead_rpv_nan.txt | grep "kernel\|fwd_full\|bwd_chain\|step" | cut -c1-200
