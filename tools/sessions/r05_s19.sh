#!/bin/bash
# round 5, session 19: whole-step suites not yet run on the build with the one-call
# stream fork (fan protocol steps, reference goldens, LDv2 steps)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 215 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fan.py tests/test_gpu_v2.py -q -x -m gpu -k "not dgrad_acc" > gpurun_out/r05s19_tests.log 2>&1; echo tests rc=$?; grep -E "passed|failed|Error" gpurun_out/r05s19_tests.log | tail -3
