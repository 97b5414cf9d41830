#!/bin/bash
# round-5 GPU session 1: the adopted straight-line trunk GEMM (one weight stream per layer, buffer loads) and the ADVICE fixes:
# the whole GPU suite, the round-4 library / the split form beside the new one in one process, the config-5 row diagnosis, a bench line
O=gpurun_out
export BN_DIAG=$PWD/$O/r05_s1_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r05_s1_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r05_s1_pytest.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
unset BN_DIAG
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py r04final BN_PP_SPLIT_BN_NO_BUFW default --config=lambert --rounds=3 > $O/r05_ab_trunk_stream_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_trunk_stream_lambert.txt | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
unset BRDFNERF_ALLOW_STALE_LIB
timeout -k 10 240 python profiles/diag_c5_rows.py --name=c5_microfacet_fp16 > $O/r05_diag_rows_c5_microfacet_fp16.txt 2>&1; rc=$?; echo "diag microfacet rc=$rc"; tail -30 $O/r05_diag_rows_c5_microfacet_fp16.txt | cut -c1-250
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 240 python profiles/diag_c5_rows.py --name=c5_hapke_theta_fp16 > $O/r05_diag_rows_c5_hapke_theta_fp16.txt 2>&1; rc=$?; echo "diag hapke rc=$rc"; tail -30 $O/r05_diag_rows_c5_hapke_theta_fp16.txt | cut -c1-250
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python bench.py > $O/r05_s1_bench_config2_bf16.json 2> $O/r05_s1_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/r05_s1_bench_config2_bf16.json')); print(round(d['value']), round(d['ms_per_step'],3), 'sustained', round(d['sustained']['value']), d['roofline']['kernel'], round(d['roofline']['frac'],3), 'per-ray', round(d['roofline']['frac_per_ray_accounting'],3), 'calib', round(d['box_calibration_before']['tflops']), round(d['box_calibration']['tflops']), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if k in ('field_fwd_full','field_bwd_chain','wgrad','skinny_wgrad')}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sweep'], d['cpu_baseline']['seconds'])"
