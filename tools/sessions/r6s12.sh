#!/bin/bash
# round 6, GPU session 12: lean ConvBnActFn (raw conv result as a bf16 C8 image, ReLU
# mask from the C8 image of z): tests, bf16 band, A/B in the step
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -m gpu -x > $O/s12_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s12_pytest.log
for rep in 1 2 3; do
echo "== bf16 default"; timeout 200 python tools/profile_step.py --mode bf16 --steps 40 --warmup 10 --pipeline 2>/dev/null | grep img/s
echo "== bf16 LD_BN_LEAN=0"; LD_BN_LEAN=0 timeout 200 python tools/profile_step.py --mode bf16 --steps 40 --warmup 10 --pipeline 2>/dev/null | grep img/s
done
