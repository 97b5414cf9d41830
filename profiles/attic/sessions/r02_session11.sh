#!/bin/bash
# round-2 GPU session 11: per-GPU step time at the strong-scaling shapes (4096 rays split over 2 / 4 / 8 GPUs)
for r in 2048 1024 512; do
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --rays $r --no-cpu-baseline > gpurun_out/bench_rays$r.json 2> gpurun_out/bench_rays$r.err || tail -3 gpurun_out/bench_rays$r.err
  python - $r <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bench_rays{sys.argv[1]}.json"))
ks = d["kernels"]
big = sum(v["ms_per_launch"] * v["launches_per_step"] for k, v in ks.items())
print(sys.argv[1], "rays:", round(d["ms_per_step"], 3), "ms/step,", round(d["value"]), "rays/s; library kernels", round(big, 3), "ms/step")
PY
done
