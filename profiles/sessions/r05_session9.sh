#!/bin/bash
# round-5 GPU session 9: re-tune of the wave priorities on the round-5 trunk stream (priority 0 / 2 while multiplying, static
# priority for the later-dispatched half) and the next layer's weight ring filled in the LAST QUARTER of the epilogue
O=gpurun_out
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py BN_GEMM_PRIO-0 BN_GEMM_PRIO-2 BN_PRIO_YOUNG BN_WRING_LATE default --config=lambert --rounds=3 > $O/r05_ab_prio_wringlate_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_prio_wringlate_lambert.txt | cut -c1-220
