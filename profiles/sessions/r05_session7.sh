#!/bin/bash
# round-5 GPU session 7: the config-5 PSNR gates at 400 and 800 BRDF-stage steps before the continuation (is 0.05 dB resolvable where the curve still climbs?)
O=gpurun_out
for n in 400 800; do
export BN_DIAG=$PWD/$O/r05_s7_c5_gate_$n.txt BN_C5_BRDF_STEPS=$n
rm -f $BN_DIAG
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tracks_fp32_config5" > $O/r05_s7_pytest_$n.log 2>&1; rc=$?; echo "pytest $n rc=$rc"; tail -3 $O/r05_s7_pytest_$n.log | cut -c1-200
cut -c1-420 $BN_DIAG
if [ $rc -ge 124 ]; then exit $rc; fi
done
