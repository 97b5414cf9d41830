#!/bin/bash
# round-2 GPU session 13: native-order Y stash vs the row-major one, same process, alternating (config 2 and config 3)
timeout -k 10 400 python profiles/ab_kernels.py prev_rowmajorY default --rounds=6 > gpurun_out/ab_nativeY.txt 2>&1 || tail -5 gpurun_out/ab_nativeY.txt
cat gpurun_out/ab_nativeY.txt
timeout -k 10 400 python profiles/ab_kernels.py prev_rowmajorY default --rounds=4 --config=rpv_nan > gpurun_out/ab_nativeY_c3.txt 2>&1 || tail -5 gpurun_out/ab_nativeY_c3.txt
cat gpurun_out/ab_nativeY_c3.txt
