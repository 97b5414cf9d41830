#!/bin/bash
# round-3 GPU session 3: full GPU suite on the lean-step build, then bench lines (config 2 at 4096 and 512 rays, config 3)
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors_s3.txt
rm -f $BN_DIAG
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r3t3.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t3.log | cut -c1-250 | head -40
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r3b3_$name.json 2> gpurun_out/r3b3_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r3b3_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3b3_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms |", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3))
print("   ", {k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items()})
PY
}
run lambert_bf16 --steps 30 --warmup 5 --no-cpu-baseline
run lambert_bf16_512 --steps 100 --warmup 10 --rays 512 --no-cpu-baseline
run lambert_fp16 --steps 30 --warmup 5 --dtype fp16 --no-cpu-baseline
run rpv_nan_bf16 --steps 20 --warmup 5 --config rpv_nan --no-cpu-baseline
