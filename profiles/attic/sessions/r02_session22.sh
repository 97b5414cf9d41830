#!/bin/bash
# round-2 GPU session 22: deterministic mode - bitwise reproducibility tests, its cost on config 2 / 3, then the full GPU suite
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "deterministic or everything_on or rccl" > gpurun_out/t22a.log 2>&1; rc=$?
tail -25 gpurun_out/t22a.log
[ $rc -eq 0 ] || exit $rc
for cfg in lambert rpv_nan; do
  timeout -k 10 200 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/det_off_$cfg.json 2> gpurun_out/det_off_$cfg.err || exit 1
  BRDFNERF_DETERMINISTIC=1 timeout -k 10 200 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/det_on_$cfg.json 2> gpurun_out/det_on_$cfg.err || exit 1
  python - $cfg <<'PY'
import json, sys
for m in ("off", "on"):
    d = json.load(open(f"gpurun_out/det_{m}_{sys.argv[1]}.json"))
    k = d["kernels"]
    print(sys.argv[1], "deterministic", m, round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms | wgrad", round(k["wgrad"]["ms_per_launch"], 3), "x", k["wgrad"]["launches_per_step"], "skinny", round(k["skinny_wgrad"]["ms_per_launch"], 3), "x", k["skinny_wgrad"]["launches_per_step"])
PY
done
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t22.log 2>&1
tail -6 gpurun_out/t22.log
