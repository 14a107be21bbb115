#!/bin/bash
# round 6, GPU session 8: BN backward without the unread fp32 copy of d(conv output)
# (bf16 mode): bit identity, bf16 suite, A/B in the step
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_graph.py -q -m gpu -x > $O/s8_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s8_pytest.log
for rep in 1 2; do
for on in 0 1; do
echo "== bf16 LD_DRAW_C8_ONLY=$on"; LD_DRAW_C8_ONLY=$on timeout 200 python tools/profile_step.py --mode bf16 --steps 30 --warmup 8 --pipeline 2>/dev/null | grep img/s
done; done
