#!/bin/bash
# MFMA utilisation / LDS / L2 counters per kernel over a few training steps and over the sigma-only inference kernel
# (one counter group per pass).   gpurun -- 'bash profiles/pmc_mfma.sh'  ->  gpurun_out/pmc_mfma_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for prog in prof_step.py prof_sigma.py; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$i -o p -- python3 $R/profiles/$prog 3 > /tmp/pm_$i.log 2>&1 || { echo "pmc $c failed"; tail -3 /tmp/pm_$i.log; }
  done
done
python3 - <<'PY' > $R/gpurun_out/pmc_mfma_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pm_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if not any(k in name for k in ("field_", "wgrad")):
            continue
        k = (name[:58], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
kern = sorted({k[0] for k in agg})
for kn in kern:
    v = {c: agg[(kn, c)][0] / agg[(kn, c)][1] for (n, c) in agg if n == kn}
    print(kn)
    for c in sorted(v):
        print(f"    {c:28s} {v[c]:16.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        print(f"    -> MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (v['GRBM_GUI_ACTIVE'] / 8):.1f} %")
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    -> LDS bank-conflict share of LDS cycles = {100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.1f} %")
    if "TCC_HIT_sum" in v:
        print(f"    -> L2 hit rate = {100 * v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.1f} %")
PY
cat $R/gpurun_out/pmc_mfma_summary.txt | grep -E "^_Z|^wgrad|->" 
