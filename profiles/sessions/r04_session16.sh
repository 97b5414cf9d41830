#!/bin/bash
# round-4 GPU session 16: barrier-free backward trunk: scheduler flags of field_bwd.hip (register-pressure trackers on / off) x
# weight-fragment prefetch depth
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 700 python profiles/ab_kernels.py default BN_BWD_DEPTH-6 bwd_notrack bwd_notrack_d4 bwd_notrack_d6 --rounds=3 > gpurun_out/r04_ab_bwd_flags.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_bwd_flags.txt | cut -c1-250
