#!/bin/bash
# round-4 GPU session 50: wgrad256's bias column sums, single stage loop: wg_always = both tiles by v_dot2c in every wave;
# wg_select = the wave's tile (wc) picked by four selects, four v_dot2c; wg_parity (timing only) = both tiles on the 16-point steps
# of the wave's parity behind a scalar branch; r04s47 = the previous form
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py r04s47 wg_always wg_select wg_parity --config=lambert --rounds=4 > gpurun_out/r04_ab_wgrad_bias2.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_wgrad_bias2.txt | cut -c1-140 | grep "kernel\|wgrad \|step\|fwd_full\|bwd_chain"
