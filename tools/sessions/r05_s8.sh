#!/bin/bash
# round 5, session 8: the full GPU suite on the current build
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/r05s8_pytest.log 2>&1; echo pytest rc=$?; tail -30 $O/r05s8_pytest.log
