#!/bin/bash
# round 5, session 16: eager bf16 step with the weight gradients on the side stream
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_bf16_wgrad_side_stream.txt
: > $O
for v in 0 1 0 1; do
  echo "== LD_WGRAD_STREAM_BF16=$v" >> $O
  LD_WGRAD_STREAM_BF16=$v timeout 300 python tools/profile_step.py --mode bf16 --pipeline --steps 40 --warmup 3 2>&1 | grep "ms/step\|Error\|error" >> $O
done

cat $O
