#!/bin/bash
# round 4, session 6: gradient outlier diagnostics, the two gradient tests, stream priorities A/B, conv traffic by kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python tools/grad_outliers.py > $O/r04s6_grad_outliers.txt 2>&1; grep -v amdgpu.ids $O/r04s6_grad_outliers.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_resnext.py -q -m gpu -s -k "gradient_elements or v2_c2_r50" > $O/r04s6_pytest.log 2>&1; echo pytest rc=$?; grep -E "vs reference|vs float64|passed|failed|Error" $O/r04s6_pytest.log | cut -c1-700
for pr in none 1 ; do
  if [ $pr = none ]; then unset LD_SIDE_STREAM_PRIORITY; else export LD_SIDE_STREAM_PRIORITY=$pr; fi
  timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/prio $pr: /"
done
unset LD_SIDE_STREAM_PRIORITY
LD_SIDE_STREAM_PRIORITY_TEACHER=1 timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/prio teacher only: /"
LD_SIDE_STREAM_PRIORITY_WGRAD=1 timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/prio wgrad only: /"
timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/prio none again: /"
