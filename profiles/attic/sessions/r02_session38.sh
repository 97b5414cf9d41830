#!/bin/bash
# round-2 GPU session 38: the C ABI from a plain host program
timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "c_abi" > gpurun_out/t38.log 2>&1; rc=$?
tail -12 gpurun_out/t38.log | cut -c1-300
exit $rc
