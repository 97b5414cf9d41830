#!/bin/bash
# round-3 GPU session 8: one backward over both passes vs two, in one process (same library)
for r in 4096 512; do
timeout -k 10 400 python profiles/ab_kernels.py default default:merge_passes=False --config=lambert --dtype=bf16 --rounds=5 --rays=$r > gpurun_out/r3ab8_merge_$r.txt 2>&1 || { echo "ab failed"; tail -5 gpurun_out/r3ab8_merge_$r.txt; }
tail -16 gpurun_out/r3ab8_merge_$r.txt
done
timeout -k 10 400 python profiles/ab_kernels.py default default:merge_passes=False --config=rpv_nan --dtype=bf16 --rounds=3 > gpurun_out/r3ab8_merge_rpv.txt 2>&1 || { echo "ab failed"; tail -5 gpurun_out/r3ab8_merge_rpv.txt; }
tail -16 gpurun_out/r3ab8_merge_rpv.txt
