#!/bin/bash
# round 6, GPU session 16: whole GPU suite on the trunk_c8_scope build
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/s16_pytest.log 2>&1; echo pytest rc=$?; tail -5 gpurun_out/s16_pytest.log
