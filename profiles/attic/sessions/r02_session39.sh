#!/bin/bash
# round-2 GPU session 39: where a small (strong-scaling) step goes - 512 and 1024 rays per GPU
for r in 512 1024 2048; do
  timeout -k 10 200 python bench.py --rays $r --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/small_$r.json 2> gpurun_out/small_$r.err || exit 1
  python - $r <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/small_{sys.argv[1]}.json"))
k = d["kernels"]
tot = sum(v["ms_per_launch"] * v["launches_per_step"] for v in k.values())
print(sys.argv[1], "rays:", round(d["ms_per_step"], 3), "ms/step; kernels sum", round(tot, 3), {n: (round(v["ms_per_launch"], 3), v["launches_per_step"]) for n, v in k.items()})
PY
done
