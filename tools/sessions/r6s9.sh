#!/bin/bash
# round 6, GPU session 9: A/B of LD_DRAW_C8_ONLY in the bf16 step, alternating
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3 4; do
for on in 1 0; do
echo "== bf16 LD_DRAW_C8_ONLY=$on"; LD_DRAW_C8_ONLY=$on timeout 200 python tools/profile_step.py --mode bf16 --steps 40 --warmup 10 --pipeline 2>/dev/null | grep img/s
done; done
