#!/bin/bash
# round-2 GPU session 7: fp16 double-backward correctness, MultiBRDF fused step, native-dZ / no-Y-copy ablations
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors_e.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests -m gpu -q -k "analytic_normals_half or fused_trainer_matches or render_image" > gpurun_out/t7.log 2>&1
tail -12 gpurun_out/t7.log
grep "analytic normals" $BN_DIAG | awk '{print $1, $7, $8, $9, $10, $11, $12, $13}' | sort | uniq -c | sort -k5 | head -50
timeout -k 10 400 python profiles/ab_kernels.py default BN_AB_BWD_NATIVE_DZ BN_AB_NO_Y_COPY --rounds=5 > gpurun_out/ab_native.txt 2>&1 || tail -5 gpurun_out/ab_native.txt
cat gpurun_out/ab_native.txt
