#!/bin/bash
export TMPDIR=/tmp
for q in 1 2; do
  export GPU_MAX_HW_QUEUES=$q
  timeout 300 python tools/profile_step.py --mode bf16 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/queues $q, one hipGraph bf16: /"
  timeout 300 python tools/profile_step.py --mode fp32 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/queues $q, one hipGraph fp32: /"
  timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/queues $q, eager fp32: /"
  timeout 300 python tools/profile_step.py --mode bf16 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/queues $q, eager bf16: /"
done
unset GPU_MAX_HW_QUEUES
for sh in 0 1 0 1; do
  LD_SHARE_SIDE_STREAMS=$sh timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/default queues, shared side stream=$sh, eager fp32: /"
done
LD_SHARE_SIDE_STREAMS=1 timeout 300 python tools/profile_step.py --mode bf16 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/default queues, shared side stream=1, eager bf16: /"
LD_SHARE_SIDE_STREAMS=0 timeout 300 python tools/profile_step.py --mode bf16 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/default queues, shared side stream=0, eager bf16: /"
