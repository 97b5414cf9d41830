#!/bin/bash
# round-2 GPU session 14: full validation of the committed state - GPU suite with the error table, smoke, bench lines of every
# BASELINE configuration, PMC passes, rocprofv3 kernel stats
set -o pipefail
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/t14.log 2>&1
rc=$?
tail -5 gpurun_out/t14.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r02_bench_$name.json 2> gpurun_out/r02_bench_$name.err || { echo "bench $name failed"; tail -3 gpurun_out/r02_bench_$name.err; return; }
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r02_bench_{sys.argv[1]}.json"))
r = d["roofline"]
print(sys.argv[1], round(d["value"]), "rays/s", round(d["ms_per_step"], 3), "ms |", r["kernel"], "alg", round(r["frac_algorithmic"], 3), "exe", round(r["frac_executed"], 3), "busy", r["mfma_busy"], "traffic", r["traffic"], "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
run config2_bf16 --steps 20 --warmup 5
run config2_fp16 --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline
run config3_rpv_nan_bf16 --steps 20 --warmup 3 --config rpv_nan --no-cpu-baseline
run config4_pergpu_rpv_nan_s128_bf16 --steps 20 --warmup 3 --config rpv_nan --rays 1024 --samples 128 --no-cpu-baseline
run config5_hapke_fp16 --steps 20 --warmup 3 --config hapke --dtype fp16 --no-cpu-baseline
run config5_microfacet_fp16 --steps 20 --warmup 3 --config microfacet --dtype fp16 --no-cpu-baseline
bash profiles/pmc_collect.sh lambert_bf16 rpv_nan_bf16 > gpurun_out/pmc_collect.log 2>&1; tail -4 gpurun_out/pmc_collect.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_l -o s -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sb_lambert.log 2>&1 || tail -5 $R/gpurun_out/sb_lambert.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_c -o s -- python3 $R/bench.py --config rpv_nan --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sb_config3.log 2>&1 || tail -5 $R/gpurun_out/sb_config3.log
cp $(find /tmp/sb_l -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_stats_bench_lambert.csv
cp $(find /tmp/sb_c -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_stats_bench_config3.csv
head -5 $R/gpurun_out/r02_stats_bench_lambert.csv | cut -c1-140
