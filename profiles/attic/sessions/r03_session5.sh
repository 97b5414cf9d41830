#!/bin/bash
# round-3 GPU session 5: lean tests + fuzz + the touched parity tests, bench at 4096 / 512 rays
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors_s5.txt
rm -f $BN_DIAG
timeout -k 10 800 python -m pytest tests/test_gpu_lean.py -m gpu -q > gpurun_out/r3t5a.log 2>&1
echo "lean+fuzz rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t5a.log | grep -v "where\|+  " | cut -c1-250 | head -30
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_size or backward or analytic" > gpurun_out/r3t5b.log 2>&1
echo "parity subset rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t5b.log | grep -v "where\|+  " | cut -c1-250 | head -30
grep "full size" $BN_DIAG | cut -c1-260
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r3b5_$name.json 2> gpurun_out/r3b5_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r3b5_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3b5_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | sustained", d["sustained"] and round(d["sustained"]["ms_per_step"], 3), "| launches", d["launches_per_step"], "|", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3))
print("   ", {k: (round(v["ms_per_launch"], 4), v["launches_per_step"]) for k, v in d["kernels"].items()})
PY
}
run lambert_bf16 --steps 30 --warmup 5 --no-cpu-baseline
run lambert_bf16_512 --steps 100 --warmup 10 --rays 512 --no-cpu-baseline
