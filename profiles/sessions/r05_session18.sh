#!/bin/bash
# round 5 session 18 (last sources): soak of the MultiBRDF variants of the lean step; rocprofv3 kernel trace of the 512-ray step
# (the strong-scaling shape: where its 0.985 ms are)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python profiles/soak.py 3000 multibrdf > gpurun_out/r05_soak_multibrdf.txt 2>&1; rc=$?; echo "soak rc=$rc"; tail -4 gpurun_out/r05_soak_multibrdf.txt | cut -c1-260
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_512 -o s -- python3 $R/bench.py --rays 512 --steps 200 --warmup 20 --no-cpu-baseline --sustained-seconds 0 > $R/gpurun_out/sb_512.log 2>&1 || tail -5 $R/gpurun_out/sb_512.log
cp $(find /tmp/sb_512 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_rocprofv3_kernel_stats_bench_512rays.csv
cut -c1-400 $R/gpurun_out/sb_512.log | tail -2
cut -d, -f1-4 $R/gpurun_out/r05_rocprofv3_kernel_stats_bench_512rays.csv | cut -c1-120 | head -24
