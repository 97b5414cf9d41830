#!/bin/bash
# round-5 GPU session 12: the forward's SIMD timeline with finer stamps (11 = hand-over signalled behind the epilogue, 12 = the next
# layer's accumulators started with its biases, 13 = own group's half of Y_l seen written): what the gap between a wave's epilogue and
# its next first MFMA consists of.  Diagnostic build only: the product object of field_fwd.hip is byte-identical with and without the stamps.
O=gpurun_out
timeout -k 10 200 python profiles/simd_timeline.py --no-build > $O/r05_simd_timeline_fine.txt 2>&1; echo "timeline rc=$?"; sed -n 5,28p $O/r05_simd_timeline_fine.txt | cut -c1-230
timeout -k 10 200 python profiles/simd_timeline.py --no-build --sigma > $O/r05_simd_timeline_fine_sigma.txt 2>&1; echo "timeline sigma rc=$?"; sed -n 5,18p $O/r05_simd_timeline_fine_sigma.txt | cut -c1-230
