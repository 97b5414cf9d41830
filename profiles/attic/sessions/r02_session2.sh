#!/bin/bash
# round-2 GPU session 2: fp16 Hapke debug + the rest of the suite
timeout -k 10 200 python profiles/debug_fp16.py c5_hapke_fp16 > gpurun_out/debug_fp16.txt 2>&1; tail -25 gpurun_out/debug_fp16.txt
timeout -k 10 200 python profiles/debug_fp16.py c5_hapke_fp16 bf16 > gpurun_out/debug_bf16.txt 2>&1; tail -12 gpurun_out/debug_bf16.txt
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors_b.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q -k "full_size or psnr or two_rank or count_nonfinite or bad_arguments" > gpurun_out/t2.log 2>&1
tail -15 gpurun_out/t2.log
