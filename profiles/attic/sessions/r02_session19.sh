#!/bin/bash
# round-2 GPU session 19: full GPU suite with the templated direction segment, per-kernel times of config 2
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t19.log 2>&1; rc=$?
tail -8 gpurun_out/t19.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python profiles/ab_kernels.py default --config=lambert --rounds=5 > gpurun_out/ab19_lambert.txt 2>&1 || exit 1
tail -8 gpurun_out/ab19_lambert.txt
timeout -k 10 300 python bench.py > gpurun_out/r02_bench_config2_bf16.json 2> gpurun_out/bench19.err || exit 1
cat gpurun_out/r02_bench_config2_bf16.json
