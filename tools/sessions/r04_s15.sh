#!/bin/bash
export TMPDIR=/tmp
for q in 2 3 4; do
  export GPU_MAX_HW_QUEUES=$q
  timeout 300 python tools/profile_step.py --mode bf16 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/queues $q, one hipGraph bf16: /"
  timeout 300 python tools/profile_step.py --mode fp32 --graph --steps 20 --warmup 3 2>&1 | grep "ms/step" | sed "s/^/queues $q, one hipGraph fp32: /"
  timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/queues $q, eager fp32: /"
done
