#!/bin/bash
# round 4, session 14: hardware-queue count vs (a) eager overlap under a process group, (b) hipGraph replay
export TMPDIR=/tmp
i=0
for q in 4 5 6 8; do
  export GPU_MAX_HW_QUEUES=$q
  i=$((i+1))
  LD_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29900 + i)) timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/queues $q, eager + process group: /"
  timeout 300 python tools/profile_step.py --mode fp32 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/queues $q, one hipGraph fp32: /"
  timeout 300 python tools/profile_step.py --mode bf16 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/queues $q, one hipGraph bf16: /"
done
