#!/bin/bash
# round-2 GPU session 1 (gpurun -- bash profiles/sessions/r02_session1.sh): full GPU suite (with the parity error table), clock probe, MFMA-shape ablation, bench
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/t1.log 2>&1
rc=$?
tail -5 gpurun_out/t1.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_CLOCK_STAMP/libbrdfnerf_hip.so timeout -k 10 120 python profiles/clock_probe.py > gpurun_out/clock_probe.txt 2>&1 || { tail -3 gpurun_out/clock_probe.txt; exit 1; }
BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/BN_CLOCK_STAMP_BN_CLOCK_STAMP_WGRAD/libbrdfnerf_hip.so timeout -k 10 120 python profiles/clock_probe.py > gpurun_out/clock_probe_wgrad.txt 2>&1 || { tail -3 gpurun_out/clock_probe_wgrad.txt; exit 1; }
cat gpurun_out/clock_probe.txt gpurun_out/clock_probe_wgrad.txt
timeout -k 10 240 python profiles/ab_kernels.py default BN_AB_MFMA16 --rounds=5 > gpurun_out/ab_mfma16.txt 2>&1 || { tail -5 gpurun_out/ab_mfma16.txt; exit 1; }
cat gpurun_out/ab_mfma16.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -5 gpurun_out/bench1.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench1.json"))
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "dtype")}, d["roofline"], d.get("cpu_baseline"))
print({k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items()})
PY
