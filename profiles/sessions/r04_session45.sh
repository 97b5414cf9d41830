#!/bin/bash
# round-4 GPU session 45: the 8-bit D stash packed by v_cvt_pknorm_i16 + v_perm (4 vector instructions per 4 values instead of 9):
# encode / decode round trip on the device, the parity tests that touch the 16-bit backward, then the A/B against the previous
# library (r04s43 = the sources of session 43)
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "d8 or backward or fused_trainer or normal or render or full_size or psnr or reproducible" > gpurun_out/r4t45.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed\|d8_roundtrip" gpurun_out/r4t45.log | cut -c1-250 | head -20
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py r04s43 default --config=lambert --rounds=4 > gpurun_out/r04_ab_d8_pknorm_lambert.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_d8_pknorm_lambert.txt | cut -c1-100
timeout -k 10 300 python profiles/ab_kernels.py r04s43 default --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_d8_pknorm_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -16 gpurun_out/r04_ab_d8_pknorm_rpv_nan.txt | cut -c1-100
