#!/bin/bash
# round 5 session 15: the fused per-sample BRDF kernel of the MultiBRDF lean step - its tests, the step time before / after
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_lean.py -x -q -m gpu -k "sample_brdf or multibrdf" > gpurun_out/s15_tests.log 2>&1 || { tail -30 gpurun_out/s15_tests.log; exit 1; }
tail -3 gpurun_out/s15_tests.log
timeout -k 10 300 python _prev/profiles/multibrdf_step.py > gpurun_out/s15_multibrdf_prev.txt 2>&1 || { tail -20 gpurun_out/s15_multibrdf_prev.txt; exit 1; }
timeout -k 10 300 python profiles/multibrdf_step.py > gpurun_out/s15_multibrdf_new.txt 2>&1 || { tail -20 gpurun_out/s15_multibrdf_new.txt; exit 1; }
cat gpurun_out/s15_multibrdf_prev.txt gpurun_out/s15_multibrdf_new.txt
timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "lean" > gpurun_out/s15_fuzz.log 2>&1 || { tail -30 gpurun_out/s15_fuzz.log; exit 1; }
tail -3 gpurun_out/s15_fuzz.log
