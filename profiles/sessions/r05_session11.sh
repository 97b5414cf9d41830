#!/bin/bash
# round-5 GPU session 11: the whole GPU suite on the committed state, with the slowest tests listed (how long does the driver's run take?)
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=25 > $O/r05_s11_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -A28 "slowest 25" $O/r05_s11_pytest.log | cut -c1-150; tail -2 $O/r05_s11_pytest.log | cut -c1-200
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
