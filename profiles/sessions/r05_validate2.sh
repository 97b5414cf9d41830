#!/bin/bash
# round-5 validation 2/3 (final sources): PMC passes, the bench lines of every BASELINE configuration, rocprofv3 kernel stats over bench.py
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r05_bench_$name.json 2> gpurun_out/r05_bench_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r05_bench_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r05_bench_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | sustained", d["sustained"] and round(d["sustained"]["ms_per_step"], 3), "| launches", d["launches_per_step"], "|", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3), "busy", r["mfma_busy"], "| traffic", r["traffic"], "| step frac", round(d.get("step_frac_of_peak", 0), 3), "| calib", round(d["box_calibration_before"]["tflops"]), round(d["box_calibration"]["tflops"]), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
bash profiles/pmc_collect.sh lambert_bf16 rpv_nan_bf16 > gpurun_out/pmc_collect_r05.log 2>&1; tail -3 gpurun_out/pmc_collect_r05.log | cut -c1-300
cd $GRAFT_REPO_ROOT
cp gpurun_out/r05_pmc.json profiles/r05_pmc.json      # so that the bench lines below carry roofline.traffic / mfma_busy
run config2_bf16
run config2_fp16 --dtype fp16 --no-cpu-baseline
run config3_rpv_nan_bf16 --config rpv_nan --no-cpu-baseline
run config4_pergpu_rpv_nan_s128_bf16 --config rpv_nan --rays 1024 --samples 128 --no-cpu-baseline
run config5_hapke_fp16 --config hapke --dtype fp16 --no-cpu-baseline
run config5_microfacet_fp16 --config microfacet --dtype fp16 --no-cpu-baseline
bash profiles/stats_bench.sh > gpurun_out/stats_bench_r05.log 2>&1; tail -8 gpurun_out/stats_bench_r05.log | cut -c1-200
