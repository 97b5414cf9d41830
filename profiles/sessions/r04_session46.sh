#!/bin/bash
# round-4 GPU session 46: the whole GPU suite on the sources with the v_cvt_pknorm D stash
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r4t46.log 2>&1; echo "gpu suite rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t46.log | cut -c1-250 | head -20
