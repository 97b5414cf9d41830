#!/bin/bash
# round-2 GPU session 12: native-order Y stash (no riding copies in the forward, swizzled native staging in the weight gradient)
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors_f.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not psnr" > gpurun_out/t12.log 2>&1
tail -6 gpurun_out/t12.log
for cfg in "lambert bf16" "rpv_nan bf16" "hapke fp16"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config $1 --dtype $2 --no-cpu-baseline > gpurun_out/bench12_$1_$2.json 2> gpurun_out/bench12_$1_$2.err || { tail -3 gpurun_out/bench12_$1_$2.err; }
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/bench12_{sys.argv[1]}_{sys.argv[2]}.json"))
    print(sys.argv[1], sys.argv[2], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms")
    print("   ", {k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items() if v["ms_per_launch"] > 0.05})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
