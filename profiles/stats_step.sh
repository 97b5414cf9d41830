#!/bin/bash
# rocprofv3 --kernel-trace --stats over a few training steps and over the sigma-only inference kernel.
#   gpurun -- 'bash profiles/stats_step.sh'  ->  gpurun_out/stats_step.csv, gpurun_out/stats_sigma.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_step -o s -- python3 $R/profiles/prof_step.py 25 > /tmp/st_step.log 2>&1 || tail -5 /tmp/st_step.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_sigma -o s -- python3 $R/profiles/prof_sigma.py > /tmp/st_sigma.log 2>&1 || tail -5 /tmp/st_sigma.log
cp $(find /tmp/st_step -name "*kernel_stats.csv" | head -1) $R/gpurun_out/stats_step.csv
cp $(find /tmp/st_sigma -name "*kernel_stats.csv" | head -1) $R/gpurun_out/stats_sigma.csv
head -12 $R/gpurun_out/stats_step.csv | cut -c1-200
