#!/bin/bash
# round-2 GPU session 52: fused step without the row-table ops (non-strict draws) - trainer / loop tests, bench
timeout -k 10 900 python -m pytest tests -m gpu -q -k "fused or train_loop or two_rank or full_size or psnr_tracks_fp32_lambert or deterministic or ray_table or guided" > gpurun_out/t52.log 2>&1; rc=$?
tail -3 gpurun_out/t52.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
for r in 4096 512; do
  timeout -k 10 200 python bench.py --rays $r --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b52.json 2> gpurun_out/b52.err || exit 1
  python - $r <<'PY'
import json, sys
d = json.load(open("gpurun_out/b52.json")); k = d["kernels"]
tot = sum(v["ms_per_launch"] * v["launches_per_step"] for v in k.values())
print(sys.argv[1], "rays:", round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms/step; kernels sum", round(tot, 3))
PY
done
