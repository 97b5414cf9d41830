#!/bin/bash
# round-5 GPU session 8: the next layer's first weight fragments prefetched behind a layer's MFMAs, ahead of its epilogue's stores
# (caller-owned ring), with the stash stores as inline asm (invisible to hipcc's vmcnt bookkeeping) and as builtins (visible)
O=gpurun_out
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py BN_WRING_PRE BN_WRING_PRE_BN_STASH_VISIBLE default --config=lambert --rounds=3 > $O/r05_ab_wring_pre_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_wring_pre_lambert.txt | cut -c1-200
