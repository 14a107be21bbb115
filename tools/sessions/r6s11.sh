#!/bin/bash
# round 6, GPU session 11: whole GPU suite on the lean GN / fused conv+GN / C8-only
# teacher tower build, then A/B in the bf16 step and one fp32 check
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > $O/s11_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s11_pytest.log
for rep in 1 2 3; do
echo "== bf16 default"; timeout 200 python tools/profile_step.py --mode bf16 --steps 40 --warmup 10 --pipeline 2>/dev/null | grep img/s
echo "== bf16 LD_FUSE_CONV_GN=0"; LD_FUSE_CONV_GN=0 timeout 200 python tools/profile_step.py --mode bf16 --steps 40 --warmup 10 --pipeline 2>/dev/null | grep img/s
done
echo "== fp32"; timeout 300 python tools/profile_step.py --mode fp32 --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
