#!/bin/bash
# round-2 GPU session 51: wgrad256 block count at small batches (512 / 1024 rays per GPU)
for r in 512 1024; do
  for v in default W2_BLOCKS-256 W2_BLOCKS-384 W2_BLOCKS-768 W2_BLOCKS-1024; do
    lib=$PWD/brdf_nerf_amd/build/$v/libbrdfnerf_hip.so; [ $v = default ] && lib=$PWD/brdf_nerf_amd/libbrdfnerf_hip.so
    BRDFNERF_HIP_LIB=$lib timeout -k 10 100 python bench.py --rays $r --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/w2.json 2> gpurun_out/w2.err || exit 1
    python - $r $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/w2.json")); k = d["kernels"]
print(sys.argv[1], "rays", sys.argv[2], "step", round(d["ms_per_step"], 3), "wgrad", round(k["wgrad"]["ms_per_launch"], 4))
PY
  done
done
