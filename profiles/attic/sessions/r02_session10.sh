#!/bin/bash
# round-2 GPU session 10: PMC passes for the bench line + rocprofv3 kernel stats of bench.py (config 2 and 3)
bash profiles/pmc_collect.sh lambert_bf16 rpv_nan_bf16 > gpurun_out/pmc_collect.log 2>&1
tail -30 gpurun_out/pmc_collect.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_l -o s -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sb_lambert.log 2>&1 || tail -5 $R/gpurun_out/sb_lambert.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_c -o s -- python3 $R/bench.py --config rpv_nan --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sb_config3.log 2>&1 || tail -5 $R/gpurun_out/sb_config3.log
cp $(find /tmp/sb_l -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_stats_bench_lambert.csv
cp $(find /tmp/sb_c -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_stats_bench_config3.csv
head -6 $R/gpurun_out/r02_stats_bench_lambert.csv | cut -c1-150
