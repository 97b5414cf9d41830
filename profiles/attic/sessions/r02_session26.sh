#!/bin/bash
# round-2 GPU session 26: per-file scheduler flags (trackers for the backward / adjoint units) - GPU suite, bench lines
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t26.log 2>&1; rc=$?
tail -4 gpurun_out/t26.log
[ $rc -eq 0 ] || exit $rc
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r02_bench_$name.json 2> gpurun_out/r02_bench_$name.err || { echo "bench $name failed"; tail -3 gpurun_out/r02_bench_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r02_bench_{sys.argv[1]}.json"))
k = d["kernels"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms |", {n: round(v["ms_per_launch"], 3) for n, v in k.items() if v["ms_per_launch"] > 0.05})
PY
}
run config2_bf16 --steps 20 --warmup 5 --no-cpu-baseline
run config2_fp16 --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline
run config3_rpv_nan_bf16 --steps 20 --warmup 3 --config rpv_nan --no-cpu-baseline
run config5_hapke_fp16 --steps 20 --warmup 3 --config hapke --dtype fp16 --no-cpu-baseline
run config5_microfacet_fp16 --steps 20 --warmup 3 --config microfacet --dtype fp16 --no-cpu-baseline
