#!/bin/bash
# round-5 validation 1/3 (final sources): the whole GPU suite with its measured errors, the default bench line, the strong-scaling
# shapes (512 / 1024 rays per GPU), the SIMD timeline of the forward
O=gpurun_out
export BN_DIAG=$PWD/$O/r05_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r05_v1_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r05_v1_pytest.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
unset BN_DIAG
timeout -k 10 200 python bench.py > $O/r05_bench_config2_bf16_prepmc.json 2> $O/r05_v1_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/r05_bench_config2_bf16_prepmc.json')); print(round(d['value']), round(d['ms_per_step'],3), 'sustained', round(d['sustained']['value']), d['roofline']['kernel'], round(d['roofline']['frac'],3), 'per-ray', round(d['roofline']['frac_per_ray_accounting'],3), 'calib', round(d['box_calibration_before']['tflops']), round(d['box_calibration']['tflops']), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if k in ('field_fwd_full','field_bwd_chain','wgrad','skinny_wgrad')}, 'cpu', round(d['cpu_baseline']['value'],1), d['cpu_baseline']['sweep'], d['cpu_baseline']['seconds'])"
export BRDFNERF_ALLOW_STALE_LIB=1
for r in 512 1024; do
timeout -k 10 200 python profiles/ab_kernels.py r04final default --config=lambert --rounds=3 --rays=$r > $O/r05_ab_strong_shape_$r.txt 2>&1; echo "ab $r rc=$?"; tail -14 $O/r05_ab_strong_shape_$r.txt | grep "kernel\|fwd_full\|bwd_chain\|wgrad \|step" | cut -c1-150
done
unset BRDFNERF_ALLOW_STALE_LIB
timeout -k 10 200 python profiles/simd_timeline.py --no-build > $O/r05_simd_timeline.txt 2>&1; echo "timeline rc=$?"; grep "layers 1-7" $O/r05_simd_timeline.txt | head -4
