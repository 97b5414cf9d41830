#!/bin/bash
# round-2 GPU session 29: TrainLoop with --beta
timeout -k 10 300 python -m pytest tests -m gpu -q -k "train_loop_every_stage" > gpurun_out/t29.log 2>&1; rc=$?
tail -15 gpurun_out/t29.log
exit $rc
