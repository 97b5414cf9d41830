#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py itself (default workload, then BASELINE config 3); summaries -> gpurun_out/
#   gpurun -- 'bash profiles/stats_bench.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_l -o s -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sb_lambert.log 2>&1 || tail -5 $R/gpurun_out/sb_lambert.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_c -o s -- python3 $R/bench.py --config rpv_nan --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sb_config3.log 2>&1 || tail -5 $R/gpurun_out/sb_config3.log
cp $(find /tmp/sb_l -name "*kernel_stats.csv" | head -1) $R/gpurun_out/stats_bench_lambert.csv
cp $(find /tmp/sb_c -name "*kernel_stats.csv" | head -1) $R/gpurun_out/stats_bench_config3.csv
head -8 $R/gpurun_out/stats_bench_lambert.csv | cut -c1-160
