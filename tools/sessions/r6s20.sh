#!/bin/bash
# round 6, GPU session 20: 16-byte C8 stores in the conv epilogues (v_permlane32_swap):
# bit identity through the existing C8 tests, whole-step tests, A/B via step list
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_layers.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -m gpu -x > $O/s20_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s20_pytest.log
for rep in 1 2 3; do
echo "== bench_step_list"; timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^eager|^pipelined_list" | cut -c1-150
done
