#!/bin/bash
# round 4, session 11: does the process group cost the stream overlap through the HW-queue mapping?
export TMPDIR=/tmp
export LD_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
i=0
for q in default 8 16 default; do
  i=$((i+1))
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  MASTER_PORT=$((29800 + i)) timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/forced collectives, GPU_MAX_HW_QUEUES=$q: /"
done
unset LD_FORCE_COLLECTIVES RANK WORLD_SIZE LOCAL_RANK
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/no process group, GPU_MAX_HW_QUEUES=$q: /"
done
