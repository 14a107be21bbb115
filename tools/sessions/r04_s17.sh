#!/bin/bash
export TMPDIR=/tmp
export LD_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
i=0
for cfg in "4 0" "4 1" "8 0" "8 1" "4 1"; do
  set -- $cfg; i=$((i+1))
  GPU_MAX_HW_QUEUES=$1 LD_SHARE_SIDE_STREAMS=$2 MASTER_PORT=$((30000 + i)) timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/process group, queues $1, shared side stream=$2: /"
done
unset LD_FORCE_COLLECTIVES RANK WORLD_SIZE LOCAL_RANK
LD_SHARE_SIDE_STREAMS=1 timeout 300 python tools/profile_step.py --mode fp32 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/no group, default queues, shared=1, one hipGraph fp32: /"
LD_SHARE_SIDE_STREAMS=1 timeout 300 python tools/profile_step.py --mode bf16 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/no group, default queues, shared=1, one hipGraph bf16: /"
LD_SHARE_SIDE_STREAMS=0 timeout 300 python tools/profile_step.py --mode bf16 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/no group, default queues, shared=0, one hipGraph bf16: /"
