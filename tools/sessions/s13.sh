#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for env in "X=1" "LD_WGRAD_STREAM=0" "LD_STUDENT_C8_ONLY=0" "LD_BN_BWD_C8=0" "LD_FUSE_CONV_BN=0" "LD_TEACHER_C8_ONLY=0" "LD_TEACHER_STREAM=0"; do
  echo "== $env"; env $env timeout 200 python tools/debug_pipelined.py bf16 2>&1 | grep -v amdgpu.ids
done > $O/s13_debug.txt 2>&1
cat $O/s13_debug.txt
