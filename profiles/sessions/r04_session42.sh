#!/bin/bash
# round-4 GPU session 42: library rebuilt in the re-created build container (same sources: hash c8bca4bd2feb3f02): smoke, the lean /
# reproducibility tests and the default bench line
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4t42_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4t42_smoke.log | cut -c1-200
timeout -k 10 400 python -m pytest tests/test_gpu_lean.py -q -m gpu -x > gpurun_out/r4t42_lean.log 2>&1; echo "lean rc=$?"; tail -2 gpurun_out/r4t42_lean.log | cut -c1-200
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_config2_bf16_recheck.json 2> gpurun_out/r4t42_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_config2_bf16_recheck.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["sustained"]["value"], d["cpu_baseline"]["value"])
PY
