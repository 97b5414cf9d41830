#!/bin/bash
# round-2 GPU session 3: fp16 analytic-normal fix, flat composite path, config-3 / config-5 bench lines
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors_c.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q -k "full_size or psnr or composite or fused_trainer or scale_free" > gpurun_out/t3.log 2>&1
tail -12 gpurun_out/t3.log
for cfg in "rpv_nan bf16" "hapke fp16" "microfacet fp16" "lambert fp16"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config $1 --dtype $2 --no-cpu-baseline > gpurun_out/bench_$1_$2.json 2> gpurun_out/bench_$1_$2.err || { tail -3 gpurun_out/bench_$1_$2.err; }
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/bench_{sys.argv[1]}_{sys.argv[2]}.json"))
    print(sys.argv[1], sys.argv[2], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac_executed"], 3), "dropped", d["dropped_nonfinite_grad_elems"])
    print("   ", {k: (round(v["ms_per_launch"], 4), round(v.get("algorithmic_GBs", 0))) for k, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
