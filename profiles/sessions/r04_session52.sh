#!/bin/bash
# round-4 GPU session 52 (run again as session 54 with the affine chain GEMM): the whole GPU suite on the final sources (v_cvt_pknorm D stash, wgrad256 bias sums by v_dot2c)
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r4t52.log 2>&1; echo "gpu suite rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t52.log | cut -c1-250 | head -20
