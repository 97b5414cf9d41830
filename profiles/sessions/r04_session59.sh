#!/bin/bash
# round-4 GPU session 59: the whole GPU suite once more on the final sources with the diagnostics file (measured parity errors)
BN_DIAG=$PWD/gpurun_out/r04_parity_errors.txt timeout -k 10 700 python -m pytest tests -q -m gpu -x > gpurun_out/r4t59.log 2>&1; echo "gpu suite rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t59.log | cut -c1-250 | head -20
wc -l gpurun_out/r04_parity_errors.txt
