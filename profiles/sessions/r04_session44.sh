#!/bin/bash
# round-4 GPU session 44: timing probes (results wrong in the probe builds): the forward's 8-bit D packing at 6 instead of 16 VALU
# operations per 8 elements (p_dnopack), and wgrad256 on v_mfma_f32_16x16x32 (two per 32x32x16 on accumulator quads: equal FLOPs,
# pipe cycles, operand reads; MI355X_MICROARCH.md DVFS give-back item 7) (p_wg16)
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 500 python profiles/ab_kernels.py default p_dnopack p_wg16 --config=lambert --rounds=4 > gpurun_out/r04_ab_dpack_wg16_probes.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_dpack_wg16_probes.txt | cut -c1-120
