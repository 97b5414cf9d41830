#!/bin/bash
# round-4 GPU session 19: per-phase cycles of wgrad256
timeout -k 10 600 python profiles/phase_timing.py -DBN_PHASE_TIMING_WGRAD > gpurun_out/r04_phase_timing_wgrad.txt 2>&1; echo "rc=$?"
tail -12 gpurun_out/r04_phase_timing_wgrad.txt
