#!/bin/bash
# round-3 GPU session 15: fold / unfold tiles with 16-byte operand loads - tests and timing
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -q -m gpu > gpurun_out/r3t15.log 2>&1; rc=$?
tail -3 gpurun_out/r3t15.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in lambert rpv_nan; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s15_$c -o s -- python3 $R/profiles/prof_step.py 12 $c bf16 > $R/gpurun_out/s15_$c.log 2>&1 || tail -5 $R/gpurun_out/s15_$c.log
cp $(find /tmp/s15_$c -name "*kernel_stats.csv" | head -1) $R/gpurun_out/s15_stats_$c.csv
cut -d, -f1-4 $R/gpurun_out/s15_stats_$c.csv | cut -c1-150 | head -18
done
