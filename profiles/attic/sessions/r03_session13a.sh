#!/bin/bash
# round-3 GPU validation, part a: full GPU suite (measured errors -> r03_parity_errors.txt), smoke, soak, N = 2 rehearsal on one GPU (gloo),
# per-GPU step times at the strong-scaling shapes
export BN_DIAG=$PWD/gpurun_out/r03_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3t13.log 2>&1
echo "pytest rc=$?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3t13.log | grep -v "where\|+  " | cut -c1-250 | head -30
unset BN_DIAG
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r3smoke13.log 2>&1; tail -2 gpurun_out/r3smoke13.log
timeout -k 10 300 python profiles/soak.py 1500 > gpurun_out/r03_soak.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03_soak.txt | cut -c1-400
BN_BENCH_SHARE_GPU=1 BN_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --rays 1024 --settle-seconds 0.3 --sustained-steps 0 > gpurun_out/r3b13_n2_rehearsal.json 2> gpurun_out/r3b13_n2_rehearsal.err || { echo "n2 rehearsal failed"; tail -20 gpurun_out/r3b13_n2_rehearsal.err; }
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3b13_n2_rehearsal.json"))
    print("n2 rehearsal:", round(d["value"]), "rays/s weak |", d["strong"])
except Exception as e:
    print("n2 rehearsal unreadable", e)
PY
for r in 512 1024 2048; do
timeout -k 10 200 python profiles/ab_kernels.py default --config=lambert --dtype=bf16 --rounds=3 --rays=$r > gpurun_out/r3ab13_strong_shape_$r.txt 2>&1 || { echo "ab failed"; tail -5 gpurun_out/r3ab13_strong_shape_$r.txt; }
tail -1 gpurun_out/r3ab13_strong_shape_$r.txt
done
