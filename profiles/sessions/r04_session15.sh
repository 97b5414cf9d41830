#!/bin/bash
# round-4 GPU session 15: per-phase cycles of the chain kernels with the barrier-free backward trunk
timeout -k 10 600 python profiles/phase_timing.py > gpurun_out/r04_phase_timing.txt 2>&1; echo "rc=$?"
tail -20 gpurun_out/r04_phase_timing.txt
