#!/bin/bash
# round 3 final validation A: full GPU suite + smoke
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --durations=12 > $O/final_pytest.log 2>&1; echo pytest rc=$?; grep -E "passed|failed" $O/final_pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1; echo smoke rc=$?; tail -1 $O/final_smoke.log
