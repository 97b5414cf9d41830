#!/bin/bash
# round-2 GPU session 9: RPV + analytic-normal PSNR gate (continuation form), N=2 rehearsal output
export BN_DIAG=$PWD/gpurun_out/r02_psnr_probe.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q -k "psnr and rpv" > gpurun_out/t9.log 2>&1
tail -3 gpurun_out/t9.log
cat $BN_DIAG
BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --steps 5 --warmup 2 --settle-seconds 0.2 --scaling strong > gpurun_out/bench_n2_gloo_strong.json 2> gpurun_out/bench_n2_gloo_strong.err || tail -5 gpurun_out/bench_n2_gloo_strong.err
python -c "
import json
d = json.load(open('gpurun_out/bench_n2_gloo_strong.json'))
print(d['n_gpus'], d['scaling'], round(d['value']), d['config']['rays_per_gpu'], d['config']['backend'])"
