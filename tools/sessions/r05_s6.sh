#!/bin/bash
# round 5, session 6: failure rate of the pipelined bf16 graph test under switches
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
: > $O/r05_stress_pipelined.txt
for v in "LD_DUMMY=1" "LD_DEFER_GRADS=0" "LD_TEACHER_REPLAY=0" "LD_FAN_FUSE=0"; do
  ( export $v; timeout 400 python tools/stress_pipelined.py 10 bf16 2>&1 | grep -v amdgpu.ids | tail -14 ) >> $O/r05_stress_pipelined.txt
done
cat $O/r05_stress_pipelined.txt
