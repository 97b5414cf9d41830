#!/bin/bash
# round-4 GPU session 40: timing probes of the forward's tail (head pass): parameter loads replaced by constants, no G / DG stash,
# no stash of the last trunk layer (results wrong in the probe builds; the question is where the tail's ~70 k cycles per tile go)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py default p_prm p_nohs p_notail p_all --config=lambert --rounds=3 > gpurun_out/r04_ab_fwd_tail_probes.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_fwd_tail_probes.txt | cut -c1-170
