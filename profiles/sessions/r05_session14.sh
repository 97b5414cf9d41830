#!/bin/bash
# round-5 GPU session 14: timing probes of the forward trunk's epilogue (results wrong; learning rate 0, finite operands in every arm):
# without its stash stores (values kept live), without the cosine + 8-bit packing (the D piece stores the Y bits)
O=gpurun_out
export BRDFNERF_ALLOW_STALE_LIB=1
timeout -k 10 300 python profiles/ab_kernels.py BN_PROBE_EPI_NOSTORE:lr=0,zero_stash=1 BN_PROBE_EPI_NOCOS:lr=0 default:lr=0 --config=lambert --rounds=3 > $O/r05_ab_epilogue_probes_lambert.txt 2>&1; rc=$?; echo "ab rc=$rc"
tail -16 $O/r05_ab_epilogue_probes_lambert.txt | cut -c1-200
