#!/bin/bash
# round-2 GPU session 50: skinny weight-gradient splits at a small batch (512 rays per GPU)
for sp in 4 8 16 32 64 128 256 512; do
  BN_SKINNY_SPLITS_DEBUG=$sp timeout -k 10 100 python bench.py --rays 512 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/sk.json 2> gpurun_out/sk.err || exit 1
  python - $sp <<'PY'
import json, sys
d = json.load(open("gpurun_out/sk.json")); k = d["kernels"]
print("splits", sys.argv[1], "step", round(d["ms_per_step"], 3), "skinny", round(k["skinny_wgrad"]["ms_per_launch"], 4), "wgrad", round(k["wgrad"]["ms_per_launch"], 4))
PY
done
