#!/bin/bash
# round 5, session 10: does a high-priority main stream help the overlapped step?
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_main_stream_priority.txt
: > $O
for m in bf16 fp32; do for p in "" "--hi-prio" ""; do
  echo "== $m $p" >> $O
  timeout 300 python tools/profile_step.py --mode $m --pipeline --steps 40 --warmup 3 $p 2>&1 | grep "ms/step\|Error\|error" >> $O
done; done
cat $O
