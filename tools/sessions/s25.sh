#!/bin/bash
# round 3: last full GPU suite + smoke on the committed build
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x --durations=8 > $O/final3_pytest.log 2>&1; echo pytest rc=$?; grep -E "passed|failed" $O/final3_pytest.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/final3_smoke.log 2>&1; echo smoke rc=$?; tail -1 $O/final3_smoke.log
