#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_graph.py tests/test_gpu_e2e.py tests/test_gpu_rccl.py -q -m gpu -x > $O/r04s20_pytest.log 2>&1; echo pytest rc=$?; grep -E "passed|failed" $O/r04s20_pytest.log | tail -1
for v in 0 1 0 1; do
  LD_BIAS_GRAD_SIDE=$v timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/bias grad on the background stream=$v: /"
done
