#!/bin/bash
# round-3 GPU session 4: full GPU suite on the lean-step build (fold / unfold / tail / adam kernels tuned), A/B of the prefetch-depth
# and young-half-priority switches in one process, bench lines
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors_s4.txt
rm -f $BN_DIAG
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r3t4.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t4.log | grep -v "where\|+  " | cut -c1-250 | head -40
timeout -k 10 400 python profiles/ab_kernels.py default BN_BWD_DEPTH-4 BN_FWD_DEPTH_TRAIN-4 BN_PRIO_YOUNG --config=lambert --dtype=bf16 --rounds=5 > gpurun_out/r3ab4_lambert_bf16.txt 2>&1 || { echo "ab failed"; tail -5 gpurun_out/r3ab4_lambert_bf16.txt; }
tail -18 gpurun_out/r3ab4_lambert_bf16.txt
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r3b4_$name.json 2> gpurun_out/r3b4_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/r3b4_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3b4_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | sustained", d["sustained"] and round(d["sustained"]["ms_per_step"], 3), "| launches", d["launches_per_step"], "|", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3))
print("   ", {k: (round(v["ms_per_launch"], 4), v["launches_per_step"]) for k, v in d["kernels"].items()})
PY
}
run lambert_bf16 --steps 30 --warmup 5 --no-cpu-baseline
run lambert_bf16_512 --steps 100 --warmup 10 --rays 512 --no-cpu-baseline
run rpv_nan_bf16 --steps 20 --warmup 5 --config rpv_nan --no-cpu-baseline
