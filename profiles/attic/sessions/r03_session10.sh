#!/bin/bash
# round-3 GPU session 10: paired PSNR study of the BRDF stage, 16 seeds
timeout -k 10 1100 python profiles/psnr_paired_study.py --seeds=16 --steps=600 > gpurun_out/r3_psnr_paired_rpv.txt 2>&1; rc=$?
tail -30 gpurun_out/r3_psnr_paired_rpv.txt
exit $rc
