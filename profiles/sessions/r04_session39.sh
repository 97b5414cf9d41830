#!/bin/bash
# round-4 GPU session 39 (run twice: first seed 225, then 249): 24 more seeds of the paired PSNR study
timeout -k 10 1150 python profiles/psnr_paired_study.py --seeds=24 --first-seed=$1 --steps=600 > gpurun_out/r04_psnr_paired_rpv_$1.txt 2>&1; echo "rc=$?"
tail -6 gpurun_out/r04_psnr_paired_rpv_$1.txt | cut -c1-200
