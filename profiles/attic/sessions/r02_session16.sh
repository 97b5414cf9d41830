#!/bin/bash
# round-2 GPU session 16: full suite after the MFMA sigma head (16-bit modes)
export BN_DIAG=$PWD/gpurun_out/r02_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/t16.log 2>&1
tail -4 gpurun_out/t16.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench16.json 2> gpurun_out/bench16.err || tail -3 gpurun_out/bench16.err
python -c "
import json
d = json.load(open('gpurun_out/bench16.json'))
print(round(d['value']), round(d['ms_per_step'], 3), {k: round(v['ms_per_launch'], 4) for k, v in d['kernels'].items() if v['ms_per_launch'] > 0.05})"
