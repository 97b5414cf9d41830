#!/bin/bash
# round-4 GPU session 43 (first seed 273: the kernels of session 43; first seed 309: session 58, the FINAL kernels of the round): 36 more seeds each of the paired PSNR study (config 3's BRDF stage)
timeout -k 10 1150 python profiles/psnr_paired_study.py --seeds=36 --first-seed=$1 --steps=600 > gpurun_out/r04_psnr_paired_rpv_$1.txt 2>&1; echo "rc=$?"
tail -6 gpurun_out/r04_psnr_paired_rpv_$1.txt | cut -c1-200
