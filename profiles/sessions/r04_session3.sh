#!/bin/bash
# round-4 GPU session 3: the chain GEMM alone (profiles/probe_gemm_rate.py): cycles per MFMA by waves per SIMD, prefetch depth, operand traffic
timeout -k 10 300 python profiles/probe_gemm_rate.py --no-build > gpurun_out/r04_probe_gemm_rate.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/r04_probe_gemm_rate.txt
