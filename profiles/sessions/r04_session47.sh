#!/bin/bash
# round-4 GPU session 47 (run again as session 55 on the final sources: + wgrad256 bias sums by v_dot2c, affine chain GEMM): validation of the sources with the v_cvt_pknorm D stash: PMC passes, bench lines of every BASELINE
# configuration, rocprofv3 kernel stats over bench.py
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r04_bench_$name.json 2> gpurun_out/r04_bench_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r04_bench_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r04_bench_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | sustained", d["sustained"] and round(d["sustained"]["ms_per_step"], 3), "| launches", d["launches_per_step"], "|", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3), "| traffic", r["traffic"], "| step frac", round(d.get("step_frac_of_peak", 0), 3), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
bash profiles/pmc_collect.sh lambert_bf16 rpv_nan_bf16 > gpurun_out/pmc_collect47.log 2>&1; tail -3 gpurun_out/pmc_collect47.log | cut -c1-300
cd $GRAFT_REPO_ROOT
cp gpurun_out/r04_pmc.json profiles/r04_pmc.json      # so that the bench lines below carry roofline.traffic / mfma_busy
run config2_bf16
run config2_fp16 --dtype fp16 --no-cpu-baseline
run config3_rpv_nan_bf16 --config rpv_nan --no-cpu-baseline
run config4_pergpu_rpv_nan_s128_bf16 --config rpv_nan --rays 1024 --samples 128 --no-cpu-baseline
run config5_hapke_fp16 --config hapke --dtype fp16 --no-cpu-baseline
run config5_microfacet_fp16 --config microfacet --dtype fp16 --no-cpu-baseline
bash profiles/stats_bench.sh > gpurun_out/stats_bench47.log 2>&1; tail -8 gpurun_out/stats_bench47.log | cut -c1-200
