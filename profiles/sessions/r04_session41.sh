#!/bin/bash
# round-4 GPU session 41: does the placement of the stash (allocation order, spacer, shift inside the allocation) change the
# fused kernels' times?  One library, nine trainers.
timeout -k 10 500 python profiles/addr_probe.py --rounds=3 > gpurun_out/r04_addr_probe.txt 2>&1; echo "probe rc=$?"
tail -12 gpurun_out/r04_addr_probe.txt | cut -c1-330
