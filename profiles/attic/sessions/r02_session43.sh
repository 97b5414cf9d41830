#!/bin/bash
# round-2 GPU session 43: per-unit prefetch depth - GPU suite and bench lines
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/t43.log 2>&1; rc=$?
tail -4 gpurun_out/t43.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 3 --config rpv_nan" "--steps 20 --warmup 5 --dtype fp16"; do
  timeout -k 10 300 python bench.py $args --no-cpu-baseline > gpurun_out/b43.json 2> gpurun_out/b43.err || exit 1
  python - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/b43.json")); k = d["kernels"]
print(sys.argv[1], "|", round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms |", {n: round(v["ms_per_launch"], 3) for n, v in k.items() if v["ms_per_launch"] > 0.05})
PY
done
