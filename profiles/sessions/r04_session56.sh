#!/bin/bash
# round-4 GPU session 56: session 55's box ran every kernel ~8 % slower than the boxes before it (forward 1.20 vs 1.10 ms with
# unchanged code): the final sources against the library of session 43 (the round's state before the pknorm D stash, the v_dot2c
# bias sums and the affine chain GEMM) in ONE process, then the bench lines of configs 2 and 3 on this box
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py r04s43 default --config=lambert --rounds=4 > gpurun_out/r04_ab_final_vs_s43_lambert.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_final_vs_s43_lambert.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_\|bwd_chain"
timeout -k 10 300 python profiles/ab_kernels.py r04s43 default --config=rpv_nan --rounds=3 > gpurun_out/r04_ab_final_vs_s43_rpv_nan.txt 2>&1; echo "ab rc=$?"
tail -16 gpurun_out/r04_ab_final_vs_s43_rpv_nan.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_\|bwd_chain\|adjoint"
unset BRDFNERF_ALLOW_STALE_LIB
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_config2_bf16_box56.json 2> gpurun_out/r04_bench_box56.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --config rpv_nan --no-cpu-baseline > gpurun_out/r04_bench_config3_rpv_nan_bf16_box56.json 2>> gpurun_out/r04_bench_box56.err; echo "bench rc=$?"
python - <<'PY'
import json
for n in ("config2_bf16", "config3_rpv_nan_bf16"):
    d = json.load(open(f"gpurun_out/r04_bench_{n}_box56.json"))
    print(n, round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), {k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items() if k in ("field_fwd_full", "field_bwd_chain", "wgrad")})
PY
