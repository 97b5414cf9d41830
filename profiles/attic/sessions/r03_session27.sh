#!/bin/bash
# round-3 GPU session 27: one ds_write_b128 per lane in the forward epilogue (permlane32_swap) vs two ds_write_b64 - parity, then in-process A/B
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "field_forward or field_backward or half_forward or golden_fp32" -x > gpurun_out/r3t27.log 2>&1; echo "parity rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t27.log | cut -c1-250 | head -10
for d in bf16 fp16; do
timeout -k 10 300 python profiles/ab_kernels.py base default --config=lambert --dtype=$d --rounds=5 > gpurun_out/r3ab27_b128_$d.txt 2>&1 || { echo "ab failed"; tail -5 gpurun_out/r3ab27_b128_$d.txt; }
tail -14 gpurun_out/r3ab27_b128_$d.txt
done
