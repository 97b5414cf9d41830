#!/bin/bash
# round-3 GPU session 11/12: paired PSNR study of the BRDF stage, 24 more seeds per call ($1 = first seed)
timeout -k 10 1150 python profiles/psnr_paired_study.py --seeds=24 --first-seed=$1 --steps=600 > gpurun_out/r3_psnr_paired_rpv_$1.txt 2>&1; rc=$?
tail -8 gpurun_out/r3_psnr_paired_rpv_$1.txt
exit $rc
