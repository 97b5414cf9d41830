#!/bin/bash
# round-2 GPU session 44: the config-3 PSNR gate three times in a row (separate processes): how stable is every number it reports
export BN_DIAG=$PWD/gpurun_out/psnr_repeat.txt
rm -f $BN_DIAG
for i in 1 2 3; do
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rpv_analytic_normals and psnr" > gpurun_out/t44_$i.log 2>&1; echo "run $i rc=$?"
done
grep "150 more" $BN_DIAG | sed 's/.*paired differences/paired differences/' | cut -c1-260
grep "600 BRDF" $BN_DIAG | sed 's/.*3 runs per mode: //' | cut -c1-330
