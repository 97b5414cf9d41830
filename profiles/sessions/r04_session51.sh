#!/bin/bash
# round-4 GPU session 51: wgrad256's bias column sums in their final form (every wave adds up half of its row group's A tiles by
# v_dot2c; the tiles numbered from the wave's own half: no select): parity, then the A/B against the previous form (r04s47) and
# the variant with selects (wg_select)
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py -q -m gpu -x -k "backward or reproducible or fused_trainer or lean_step or full_size" > gpurun_out/r4t51.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t51.log | cut -c1-250 | head
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py r04s47 wg_select default --config=lambert --rounds=4 > gpurun_out/r04_ab_wgrad_bias3.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_wgrad_bias3.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_full\|bwd_chain"
timeout -k 10 300 python profiles/ab_kernels.py r04s47 default --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_wgrad_bias3_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -16 gpurun_out/r04_ab_wgrad_bias3_rpv_nan.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_full\|bwd_chain\|adjoint"
