#!/bin/bash
# round-5 PSNR studies on the frozen kernel sources.  usage: r05_psnr_study.sh <protocol> <first-seed> <seeds> <config> [<config> ...]
# (one gpurun call is at most 20 minutes: restart protocol ~26 s per seed, continue protocol ~3.5 s per seed + ~15 s of shared training)
O=gpurun_out
proto=$1; s0=$2; n=$3; shift 3
for cfg in "$@"; do
  out=$O/r05_psnr_${proto}_${cfg}_seeds${s0}_$((s0 + n - 1)).txt
  timeout -k 10 1150 python profiles/psnr_paired_study.py --config=$cfg --protocol=$proto --first-seed=$s0 --seeds=$n > $out 2>&1; rc=$?
  echo "$cfg $proto rc=$rc"; tail -4 $out | cut -c1-250
  if [ $rc -ge 124 ]; then exit $rc; fi
done
