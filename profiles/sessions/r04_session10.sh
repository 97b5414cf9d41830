#!/bin/bash
# round-4 GPU session 10: validation of the tree as session 9 left it: full GPU suite, PMC passes (config 2), headline bench line,
# rocprofv3 kernel stats over bench.py
export BN_DIAG=$PWD/gpurun_out/r04_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r4t10.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t10.log | grep -v "where\|+  " | cut -c1-250 | head -30
unset BN_DIAG
bash profiles/pmc_collect.sh lambert_bf16 > gpurun_out/pmc_collect10.log 2>&1; tail -3 gpurun_out/pmc_collect10.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/r04_pmc.json profiles/r04_pmc.json
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_config2_bf16.json 2> gpurun_out/r04_bench_config2_bf16.err || tail -5 gpurun_out/r04_bench_config2_bf16.err
cut -c1-1500 gpurun_out/r04_bench_config2_bf16.json
bash profiles/stats_bench.sh > gpurun_out/stats_bench10.log 2>&1; tail -8 gpurun_out/stats_bench10.log | cut -c1-200
