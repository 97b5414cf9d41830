#!/bin/bash
# round-4 GPU session 49: wgrad256's bias column sums: r04s47 = every wave, shifts / masks / adds (the sources of session 47);
# default = three stage-loop instances (none / tile 0 / tile 1, v_dot2c sums); wg_always = one loop, both tiles by v_dot2c in every
# wave; wg_two = two instances (none / the wave's tile picked by selects)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or reproducible" > gpurun_out/r4t49.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t49.log | cut -c1-250 | head
timeout -k 10 400 python profiles/ab_kernels.py r04s47 default wg_always wg_two --config=lambert --rounds=4 > gpurun_out/r04_ab_wgrad_bias.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_wgrad_bias.txt | cut -c1-140
