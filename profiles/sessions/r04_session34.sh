#!/bin/bash
# round-4 GPU session 34: the lean fuzz with the sun pass and MultiBRDF drawn (and at twice its length), then the full GPU suite
BN_FUZZ_SCALE=2 timeout -k 10 800 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "lean" > gpurun_out/r4t34.log 2>&1; echo "fuzz-lean rc=$?"
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r4t34.log | cut -c1-300 | head
