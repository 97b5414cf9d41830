#!/bin/bash
# round-4 GPU session 38: paired PSNR study of config 3's BRDF stage on the round-4 kernels (24 seeds x {fp32, bf16, fp16})
timeout -k 10 1150 python profiles/psnr_paired_study.py --seeds=24 --first-seed=201 --steps=600 > gpurun_out/r04_psnr_paired_rpv.txt 2>&1; echo "rc=$?"
tail -10 gpurun_out/r04_psnr_paired_rpv.txt | cut -c1-200
