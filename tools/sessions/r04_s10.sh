#!/bin/bash
# round 4, session 10: which collective costs what in the forced 1-rank group
export TMPDIR=/tmp
export LD_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
i=0
for skip in "" buckets norm logs "buckets,norm,logs" ""; do
  i=$((i+1))
  MASTER_PORT=$((29700 + i)) LD_COLLECTIVES_SKIP=$skip timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/skip [$skip]: /"
done
