#!/bin/bash
# round-4 GPU session 62: one box, three numbers: the box-speed indicator (bench.py box_calibration), the final sources beside the
# library of session 43 in one process, and the full default bench line of the final sources
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 120 python profiles/ab_kernels.py r04s43 default --config=lambert --rounds=3 > gpurun_out/r04_ab_final_vs_s43_lambert_box62.txt 2>&1; echo "ab rc=$?"
tail -13 gpurun_out/r04_ab_final_vs_s43_lambert_box62.txt | cut -c1-110 | grep "kernel\|wgrad \|step\|fwd_full\|bwd_chain"
unset BRDFNERF_ALLOW_STALE_LIB
timeout -k 10 200 python bench.py > gpurun_out/r04_bench_config2_bf16_box62.json 2> gpurun_out/r04_bench_box62.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_config2_bf16_box62.json')); print(round(d['value']), round(d['ms_per_step'],3), 'sustained', round(d['sustained']['value']), d['roofline']['kernel'], round(d['roofline']['frac'],3), 'calib', round(d['box_calibration']['tflops']), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if k in ('field_fwd_full','field_bwd_chain','wgrad')}, 'cpu', d['cpu_baseline']['value'])"
