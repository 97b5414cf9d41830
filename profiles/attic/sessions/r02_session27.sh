#!/bin/bash
# round-2 GPU session 27: study of the config-3 PSNR gate over starting states and continuation set-ups
timeout -k 10 900 python profiles/psnr_state_study.py 4 > gpurun_out/r02_psnr_state_study.txt 2>&1
tail -20 gpurun_out/r02_psnr_state_study.txt
