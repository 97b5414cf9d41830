#!/bin/bash
# round-3 GPU session 6: full GPU suite + smoke on the launch-lean build, the N = 2 bench path rehearsed on one GPU (gloo),
# bench lines of every BASELINE configuration
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3t6.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t6.log | grep -v "where\|+  " | cut -c1-250 | head -30
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r3smoke.log 2>&1; tail -2 gpurun_out/r3smoke.log
BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --rays 1024 --settle-seconds 0.3 --sustained-steps 0 > gpurun_out/r3b6_n2_rehearsal.json 2> gpurun_out/r3b6_n2_rehearsal.err || { echo "n2 rehearsal failed"; tail -20 gpurun_out/r3b6_n2_rehearsal.err; }
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3b6_n2_rehearsal.json"))
    print("n2 rehearsal:", round(d["value"]), "rays/s weak |", d["strong"])
except Exception as e:
    print("n2 rehearsal unreadable", e)
PY
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r3b6_$name.json 2> gpurun_out/r3b6_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r3b6_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3b6_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | sustained", d["sustained"] and round(d["sustained"]["ms_per_step"], 3), "| launches", d["launches_per_step"], "|", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
run config2_bf16
run config2_fp16 --dtype fp16 --no-cpu-baseline
run config3_rpv_nan_bf16 --config rpv_nan --no-cpu-baseline
run config4_pergpu_rpv_nan_s128_bf16 --config rpv_nan --rays 1024 --samples 128 --no-cpu-baseline
run config5_hapke_fp16 --config hapke --dtype fp16 --no-cpu-baseline
run config5_microfacet_fp16 --config microfacet --dtype fp16 --no-cpu-baseline
