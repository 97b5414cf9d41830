#!/bin/bash
# round-4 GPU session 11: backward-chain probes: riding stores confined to one half of the k-loop per wave (SIMD partners in
# opposite halves), and the timing-only probes without the riding stores / without the derivative loads
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 600 python profiles/ab_kernels.py default BN_BWD_HALF_STORES BN_PROBE_NO_RIDE BN_PROBE_NO_D BN_PROBE_NO_RIDE_BN_PROBE_NO_D --rounds=3 > gpurun_out/r04_ab_bwd_probes.txt 2>&1; echo "ab rc=$?"
tail -22 gpurun_out/r04_ab_bwd_probes.txt | cut -c1-220
