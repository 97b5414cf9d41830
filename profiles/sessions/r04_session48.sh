#!/bin/bash
# round-4 GPU session 48: timing probe: wgrad256 without the bias column sums every wave adds up in its k-loop (~36 vector
# instructions per 8 MFMAs, a quarter of the waves store them)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py default p_wgnobias --config=lambert --rounds=4 > gpurun_out/r04_ab_wgrad_nobias_probe.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_wgrad_nobias_probe.txt | cut -c1-100
