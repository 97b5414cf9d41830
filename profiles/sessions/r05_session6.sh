#!/bin/bash
# round-5 GPU session 6: the whole GPU suite on the frozen kernel sources (fp16 derivative stash, adjoint stream, config-5 PSNR gates)
O=gpurun_out
export BN_DIAG=$PWD/$O/r05_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_s6_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r05_s6_pytest.log | cut -c1-300
grep "config 5" $BN_DIAG | cut -c1-400
