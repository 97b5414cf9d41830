#!/bin/bash
# round-5 GPU session 5: wgrad256 split into a bias / no-bias instantiation (two launches of one grid): parity subset, then A/B against
# the one-launch form and the prefetch depths 8 (forward) / 4 (backward trunk); config 3: the adjoint chain as a straight-line stream
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py -m gpu -x -q -k "backward or wgrad or reproducible or lean_step_matches or full_width" > $O/r05_s5_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r05_s5_pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python profiles/ab_kernels.py BN_WGRAD_BIAS_INLINE BN_FWD_DEPTH_TRAIN-8_BN_BWD_PP_DEPTH-4 default --config=lambert --rounds=3 > $O/r05_ab_wgrad_bias_split_depths_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_wgrad_bias_split_depths_lambert.txt | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python profiles/ab_kernels.py BN_ADJ_FIXED default --config=rpv_nan --rounds=3 > $O/r05_ab_adj_fixed_rpv_nan.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -20 $O/r05_ab_adj_fixed_rpv_nan.txt | cut -c1-200
