#!/bin/bash
# round-5 validation 3/3 (final sources): soak, evaluation throughput, the N > 1 bench path rehearsed with six ranks on one GPU (gloo)
timeout -k 10 700 python profiles/soak.py 3000 > gpurun_out/r05_soak.txt 2>&1; rc=$?; echo "soak rc=$rc"
tail -12 gpurun_out/r05_soak.txt | cut -c1-220
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python profiles/eval_throughput.py > gpurun_out/r05_eval_throughput.txt 2>&1; echo "eval rc=$?"
tail -8 gpurun_out/r05_eval_throughput.txt | cut -c1-200
BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 6 --rays 3072 --steps 10 --warmup 2 --no-cpu-baseline --sustained-seconds 1 > gpurun_out/r05_bench_n6_rehearsal_one_gpu_gloo.json 2> gpurun_out/r05_bench_n6_rehearsal.err || tail -5 gpurun_out/r05_bench_n6_rehearsal.err
cut -c1-700 gpurun_out/r05_bench_n6_rehearsal_one_gpu_gloo.json
