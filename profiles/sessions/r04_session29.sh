#!/bin/bash
# round-4 GPU session 29: backward-trunk prefetch depth 4 / 8 against 6 once more on the final kernel; bench lines of configs 3-5
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 400 python profiles/ab_kernels.py default BN_BWD_PP_DEPTH-4 BN_BWD_PP_DEPTH-8 --rounds=3 > gpurun_out/r04_ab_bwd_pp_depth.txt 2>&1; echo "ab rc=$?"
tail -14 gpurun_out/r04_ab_bwd_pp_depth.txt | cut -c1-120 | grep -v "pack\|composite\|guided\|strat\|adam\|skinny\|reduce"
unset BRDFNERF_ALLOW_STALE_LIB
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r04_bench_$name.json 2> gpurun_out/r04_bench_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r04_bench_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r04_bench_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | sustained", d["sustained"] and round(d["sustained"]["ms_per_step"], 3), "| launches", d["launches_per_step"], "|", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "| step frac", round(d.get("step_frac_of_peak", 0), 3))
PY
}
run config3_rpv_nan_bf16 --config rpv_nan --no-cpu-baseline
run config4_pergpu_rpv_nan_s128_bf16 --config rpv_nan --rays 1024 --samples 128 --no-cpu-baseline
run config5_hapke_fp16 --config hapke --dtype fp16 --no-cpu-baseline
run config5_microfacet_fp16 --config microfacet --dtype fp16 --no-cpu-baseline
