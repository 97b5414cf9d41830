#!/bin/bash
# round-5 GPU session 4: the fp16 derivative stash (DK16) of the fp16 mode with analytic normals: parity tests that touch it, and the
# row diagnosis of config 5 again (does the whole-gradient cosine recover?)
O=gpurun_out
export BN_DIAG=$PWD/$O/r05_s4_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or normals or analytic or 16bit or fp16 or half" > $O/r05_s4_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/r05_s4_pytest.log | cut -c1-300
grep "END-TO-END" $BN_DIAG | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
unset BN_DIAG
timeout -k 10 240 python profiles/diag_c5_rows.py --name=c5_microfacet_fp16 > $O/r05_diag_rows_c5_microfacet_fp16_dk16.txt 2>&1; rc=$?; echo "diag microfacet rc=$rc"; grep "whole flat\|analytic-normal angle\|^---\|gradient rows" $O/r05_diag_rows_c5_microfacet_fp16_dk16.txt | cut -c1-220
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 240 python profiles/diag_c5_rows.py --name=c5_hapke_theta_fp16 > $O/r05_diag_rows_c5_hapke_theta_fp16_dk16.txt 2>&1; rc=$?; echo "diag hapke rc=$rc"; grep "whole flat\|analytic-normal angle\|^---\|gradient rows" $O/r05_diag_rows_c5_hapke_theta_fp16_dk16.txt | cut -c1-220
