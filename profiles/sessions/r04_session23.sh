#!/bin/bash
# round-4 GPU session 23: strong-scaling shapes (512 / 1024 rays per GPU), session 9's library against the current one
export BRDFNERF_ALLOW_STALE_LIB=1
for r in 512 1024; do
timeout -k 10 300 python profiles/ab_kernels.py r04s9 default --rays=$r --rounds=3 > gpurun_out/r04_ab_strong_shape_$r.txt 2>&1; echo "ab $r rc=$?"
tail -14 gpurun_out/r04_ab_strong_shape_$r.txt | cut -c1-120
done
