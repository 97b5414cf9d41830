#!/bin/bash
# round-4 GPU session 31: timing probe: wgrad256 with the native A chunks written to LDS without the lane-pair swap
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py default W2_PROBE_NO_SWAP_A --rounds=4 > gpurun_out/r04_ab_wgrad_noswap_probe.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_wgrad_noswap_probe.txt | cut -c1-100 | grep -v "pack\|composite\|guided\|strat\|adam\|skinny\|reduce"
