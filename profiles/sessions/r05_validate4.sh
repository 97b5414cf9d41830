#!/bin/bash
# round-5 final pass on the final sources: the whole GPU suite with its measured errors, then validate2 (PMC passes, the bench lines
# of every BASELINE configuration, rocprofv3 kernel stats over bench.py)
O=gpurun_out
export BN_DIAG=$PWD/$O/r05_parity_errors.txt
rm -f $BN_DIAG
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r05_v4_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r05_v4_pytest.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
unset BN_DIAG
bash profiles/sessions/r05_validate2.sh
