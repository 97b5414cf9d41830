#!/bin/bash
# rocprofv3 PMC passes over a few training steps (one counter group per pass), summarised per kernel.
#   gpurun -- 'bash profiles/pmc_step.sh'   ->  gpurun_out/pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 $R/profiles/prof_step.py 3 > /tmp/pmc_$tag.log 2>&1 || { echo "pmc $c failed"; tail -5 /tmp/pmc_$tag.log; }
done
python3 - <<'PY' > $R/gpurun_out/pmc_summary.txt
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
        for (kn, cn), (v, n) in sorted(agg.items()):
            if v / n > 1e3: print(f"{cn:28s} {kn:62s} launches {n:4d}  mean {v / n:14.1f}")
PY
cat $R/gpurun_out/pmc_summary.txt | grep -E "field_|wgrad" | head -60
