#!/bin/bash
# round-2 GPU session 31: F = 192 analytic-normal fix - shape probe, fuzz (more seeds), GPU suite
timeout -k 10 300 python profiles/debug_an_shapes.py 2>&1 | tail -5
export BN_DIAG=$PWD/gpurun_out/r02_fuzz_errors.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/t31a.log 2>&1; rc=$?
tail -8 gpurun_out/t31a.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t31.log 2>&1
tail -5 gpurun_out/t31.log
