#!/bin/bash
# round 5, session 17: eager bf16 with the weight gradients on the side stream by
# default (one-call stream fork): timing, then the suites that exercise the bf16 /
# deferred / process-group paths
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_bf16_wgrad_side_stream_fastfork.txt
: > $O
for v in 1 0 1; do
  echo "== LD_WGRAD_STREAM_BF16=$v (ld_stream_fork path)" >> $O
  LD_WGRAD_STREAM_BF16=$v timeout 200 python tools/profile_step.py --mode bf16 --pipeline --steps 40 --warmup 3 2>&1 | grep "ms/step\|Error\|error" >> $O
done
echo "== fp32 (ld_stream_fork path)" >> $O
timeout 200 python tools/profile_step.py --mode fp32 --pipeline --steps 30 --warmup 3 2>&1 | grep "ms/step\|Error\|error" >> $O
cat $O
timeout 420 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_graph.py tests/test_gpu_defer.py tests/test_gpu_graph_pg.py tests/test_gpu_rccl.py tests/test_gpu_teacher_replay.py -q -x -m gpu > gpurun_out/r05s17_tests.log 2>&1; echo tests rc=$?; grep -E "passed|failed|Error" gpurun_out/r05s17_tests.log | tail -3
