#!/bin/bash
# round-2 GPU session 48: the final tree - GPU suite exactly as the driver runs it (-x -q -m gpu), smoke, default bench line
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/t48.log 2>&1; rc=$?
tail -4 gpurun_out/t48.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/bench_default.err || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_default.json"))
r = d["roofline"]
print(round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms |", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3), "busy", r["mfma_busy"], "traffic", r["traffic"], "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
