#!/bin/bash
# round-5 GPU session 2: (a) quick parity of the forward with straight-line head / sigma GEMMs; (b) A/B: looped head GEMMs, the
# 8-bit Y timing probe (results wrong: what would halving the Y stash buy?); (c) 8-bit D codec in the fp32 mode: what the D stash
# alone does to the analytic normals of config 5
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward or backward or normals or render_golden or field" > $O/r05_s2_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r05_s2_pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python profiles/ab_kernels.py BN_NO_FIXED_FULL BN_PROBE_Y8 default --config=lambert --rounds=3 > $O/r05_ab_fixedfull_y8probe_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_fixedfull_y8probe_lambert.txt | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python profiles/diag_c5_rows.py --name=c5_microfacet_fp16 --d8lib=brdf_nerf_amd/build/BN_DIAG_D8_IN_F32/libbrdfnerf_hip.so > $O/r05_diag_rows_c5_microfacet_fp16_d8.txt 2>&1; rc=$?; echo "diag microfacet rc=$rc"; grep "whole flat\|analytic-normal angle\|^---\|gradient rows" $O/r05_diag_rows_c5_microfacet_fp16_d8.txt | cut -c1-220
