#!/bin/bash
# round-3 final validation, part b: PMC passes, bench lines of every BASELINE configuration, rocprofv3 kernel stats over bench.py,
# evaluation throughput
bash profiles/sessions/r03_session13b.sh
timeout -k 10 300 python profiles/eval_throughput.py > gpurun_out/r03_eval_throughput.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03_eval_throughput.txt | tail -12
