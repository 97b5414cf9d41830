#!/bin/bash
# round 5 session 16: per-sample irradiance of the sun pass in the per-sample shading launch (MultiBRDF + sun pass, Lambertian rgb +
# sun pass on the lean path); the lean fuzz at 4x seeds; then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_lean.py -x -q -m gpu -k "sample_brdf or multibrdf or sun_visibility" > gpurun_out/s16_tests.log 2>&1 || { tail -40 gpurun_out/s16_tests.log; exit 1; }
tail -3 gpurun_out/s16_tests.log
BN_FUZZ_SCALE=4 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "lean_step" > gpurun_out/s16_fuzz.log 2>&1 || { tail -40 gpurun_out/s16_fuzz.log; exit 1; }
tail -3 gpurun_out/s16_fuzz.log
BN_DIAG=gpurun_out/s16_parity_errors.txt timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s16_suite.log 2>&1 || { tail -40 gpurun_out/s16_suite.log; exit 1; }
tail -3 gpurun_out/s16_suite.log
