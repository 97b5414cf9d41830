#!/bin/bash
# round-2 GPU session 4: reworked PSNR gate (2 repeats), generalised flat composite, fault word, hapke fp16 bench
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors_d.txt
rm -f $BN_DIAG
timeout -k 10 1100 python -m pytest tests -m gpu -q -k "psnr or composite or fault_word or render_rays or fused_trainer" > gpurun_out/t4.log 2>&1
tail -12 gpurun_out/t4.log
grep -i psnr $BN_DIAG
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config hapke --dtype fp16 --no-cpu-baseline > gpurun_out/bench_hapke_fp16.json 2> gpurun_out/bench_hapke_fp16.err || tail -3 gpurun_out/bench_hapke_fp16.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_hapke_fp16.json"))
print("hapke fp16", round(d["value"]), "rays/s", {k: (round(v["ms_per_launch"], 4), round(v.get("algorithmic_GBs", 0))) for k, v in d["kernels"].items() if "composite" in k})
PY
