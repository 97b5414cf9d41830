#!/bin/bash
# round-2 GPU session 8: s'-free adjoint chains, PSNR gates, N=2 bench rehearsal (one GPU, gloo)
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/t8.log 2>&1
tail -8 gpurun_out/t8.log
grep -i "psnr" $BN_DIAG
BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --steps 5 --warmup 2 --settle-seconds 0.2 > gpurun_out/bench_n2_gloo.json 2> gpurun_out/bench_n2_gloo.err || tail -5 gpurun_out/bench_n2_gloo.err
BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --steps 5 --warmup 2 --settle-seconds 0.2 --scaling strong > gpurun_out/bench_n2_gloo_strong.json 2> gpurun_out/bench_n2_gloo_strong.err || tail -5 gpurun_out/bench_n2_gloo_strong.err
python - <<'PY'
import json
for f in ("bench_n2_gloo", "bench_n2_gloo_strong"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["n_gpus"], d["scaling"], round(d["value"]), d["config"]["rays_per_gpu"], d["config"]["backend"])
    except Exception as e:
        print(f, "failed", e)
PY
