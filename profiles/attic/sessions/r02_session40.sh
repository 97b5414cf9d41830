#!/bin/bash
# round-2 GPU session 40: the N > 1 bench path end to end on the final kernels (2 ranks sharing the one GPU, gloo): plumbing check
export HSA_ENABLE_IPC_MODE_LEGACY=0 BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo
for sc in weak strong; do
  timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --scaling $sc --no-cpu-baseline > gpurun_out/r02_bench_n2_rehearsal_one_gpu_gloo_$sc.json 2> gpurun_out/n2_$sc.err || { tail -5 gpurun_out/n2_$sc.err; exit 1; }
  python - $sc <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r02_bench_n2_rehearsal_one_gpu_gloo_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], {k: d[k] for k in ("value", "n_gpus", "steps", "ms_per_step", "scaling")}, d["config"].get("backend"), d["config"].get("parallelism"), d["config"].get("rays_per_gpu"))
PY
done
