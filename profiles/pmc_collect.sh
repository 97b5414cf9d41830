#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass, --kernel-trace only) over a few training steps of the bench workloads,
# summarised per kernel into gpurun_out/r05_pmc.json, which bench.py reads (copy it to profiles/r05_pmc.json) for
# roofline.traffic / roofline.mfma_busy.  The record carries a hash of csrc/: bench.py ignores it once the kernels change.
#   gpurun -- 'bash profiles/pmc_collect.sh [workload ...]'     workload = <config>_<dtype>, default: lambert_bf16 rpv_nan_bf16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
WL="${@:-lambert_bf16 rpv_nan_bf16}"
rm -rf /tmp/pmcc_*
for wl in $WL; do
  cfg=${wl%_*}; dt=${wl##*_}
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcc_${wl}_$i -o p -- python3 $R/profiles/prof_step.py 3 $cfg $dt > /tmp/pmcc_${wl}_$i.log 2>&1 \
      || { echo "pmc pass '$c' of $wl failed"; tail -5 /tmp/pmcc_${wl}_$i.log; }
    echo "pass $i of $wl done" >> $R/gpurun_out/pmc_collect.progress
  done
done
python3 $R/profiles/pmc_parse.py /tmp $R/gpurun_out/r05_pmc.json $WL
