#!/bin/bash
# round 3, session 21: after the raw-output fix -- conv / bf16 / fused / e2e tests, then
# eager and hipGraph step times in both modes
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bf16.py tests/test_gpu_e2e.py tests/test_gpu_graph.py -x -q > $O/s21_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s21_pytest.log
timeout 200 python tools/profile_step.py --mode fp32 --steps 15 --warmup 4 --pipeline > $O/s21_step_fp32.log 2>&1; echo "fp32 eager: $(grep 'ms/step' $O/s21_step_fp32.log)"
timeout 200 python tools/profile_step.py --mode fp32 --steps 15 --warmup 4 --graph > $O/s21_step_fp32_graph.log 2>&1; echo "fp32 graph: $(grep 'ms/step' $O/s21_step_fp32_graph.log)"
timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline > $O/s21_step_bf16.log 2>&1; echo "bf16 eager: $(grep 'ms/step' $O/s21_step_bf16.log)"
timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --graph > $O/s21_step_bf16_graph.log 2>&1; echo "bf16 graph: $(grep 'ms/step' $O/s21_step_bf16_graph.log)"
timeout 200 python tools/host_bound.py bf16 > $O/s21_host.log 2>&1 || true; tail -3 $O/s21_host.log
