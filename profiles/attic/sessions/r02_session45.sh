#!/bin/bash
# round-2 GPU session 45: where the 16-bit modes fall behind in the BRDF stage of config 3
timeout -k 10 1100 python profiles/psnr_transient_study.py 5 > gpurun_out/r02_psnr_transient_study.txt 2>&1
tail -20 gpurun_out/r02_psnr_transient_study.txt
