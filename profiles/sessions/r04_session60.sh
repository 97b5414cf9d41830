#!/bin/bash
# round-4 GPU session 60: the 8-GPU strong-scaling shape (512 rays per GPU) on the final sources beside the library of session 43
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 200 python profiles/ab_kernels.py r04s43 default --config=lambert --rays=512 --rounds=4 > gpurun_out/r04_ab_strong_shape_512_final.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_strong_shape_512_final.txt | cut -c1-110
