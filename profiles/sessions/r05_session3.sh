#!/bin/bash
# round-5 GPU session 3: the 8-bit Y timing probe again with the learning rate at 0 in both arms (session 2's probe arm trained on
# its own wrong weight gradients: NaN weights, a chip multiplying NaNs draws less power and clocks higher - every kernel "faster")
O=gpurun_out
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py BN_PROBE_Y8:lr=0 default:lr=0 --config=lambert --rounds=3 > $O/r05_ab_y8probe_lr0_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_y8probe_lr0_lambert.txt | cut -c1-200
