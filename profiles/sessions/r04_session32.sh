#!/bin/bash
# round-4 GPU session 32: soak (3000 / 1000 graph-replayed steps per (config, dtype)) and evaluation throughput on the final kernels
timeout -k 10 900 python profiles/soak.py 3000 > gpurun_out/r04_soak.txt 2>&1; echo "soak rc=$?"
tail -12 gpurun_out/r04_soak.txt | cut -c1-220
timeout -k 10 300 python profiles/eval_throughput.py > gpurun_out/r04_eval_throughput.txt 2>&1; echo "eval rc=$?"
tail -8 gpurun_out/r04_eval_throughput.txt | cut -c1-200
