#!/bin/bash
# round 5, session 5: bisect the NaN of test_pipelined_graphed_step_equals_eager[bf16]
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
T="tests/test_gpu_graph.py::test_pipelined_graphed_step_equals_eager"
run() { echo "== $*"; ( export "$@"; timeout 200 python -m pytest "$T" -q -x -m gpu 2>&1 | tail -3 ); }
run LD_DUMMY=1
run LD_DUMMY=2
run LD_FAN_INPLACE=0
run LD_FUSED_BOTTLENECK=0
run LD_DEFER_GRADS=0
run LD_TEACHER_REPLAY=0
run LD_FAN_FUSE=0
timeout 300 python -m pytest tests/test_gpu_fan.py tests/test_gpu_defer.py -q -x -m gpu 2>&1 | tail -3
