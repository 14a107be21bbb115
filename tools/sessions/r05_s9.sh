#!/bin/bash
# round 5, session 9: bf16 step time vs the split-cost constant of the C8
# weight-gradient heuristic (LD_WGRAD_C8_FIXED; default 6)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_wgrad_c8_split_cost.txt
: > $O
for F in 6 10 16 24 6; do
  echo "== LD_WGRAD_C8_FIXED=$F" >> $O
  LD_WGRAD_C8_FIXED=$F timeout 300 python tools/profile_step.py --mode bf16 --pipeline --steps 40 --warmup 3 2>&1 | grep "ms/step" >> $O
done
cat $O
