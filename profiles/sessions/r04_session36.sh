#!/bin/bash
# round-4 GPU session 36: what the driver runs at round end: build(), smoke(), the default bench line
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r04_smoke.log | cut -c1-200
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_default.json"))
r = d["roofline"]
print(round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms |", r["kernel"], "frac", round(r["frac"], 3), "traffic", r["traffic"], "mfma_busy", r.get("mfma_busy"), "| cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
