#!/bin/bash
# round-3 GPU session 9: ray-level shading + loss kernel (bn_ray_shade_loss) - unit test, lean-vs-general step, BRDF step timing
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_lean.py -q -m gpu > gpurun_out/r3s9_lean.log 2>&1; rc=$?
tail -15 gpurun_out/r3s9_lean.log
[ $rc -eq 0 ] || exit $rc
for c in rpv_nan hapke microfacet; do
timeout -k 10 400 python profiles/ab_kernels.py default --config=$c --dtype=bf16 --rounds=3 > gpurun_out/r3ab9_$c.txt 2>&1 || { echo "ab failed"; tail -5 gpurun_out/r3ab9_$c.txt; exit 1; }
tail -18 gpurun_out/r3ab9_$c.txt
done
