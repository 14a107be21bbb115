#!/bin/bash
# round 6, GPU session 19: GroupNorm on the C8-only conv result: tests, band, A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bf16.py tests/test_gpu_graph.py tests/test_gpu_e2e.py tests/test_gpu_v2.py -q -m gpu -x > $O/s19_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s19_pytest.log
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s -k "golden" > $O/s19_band.log 2>&1; grep -E "grad-norm relative|passed|failed" $O/s19_band.log
for rep in 1 2 3; do
for cfg in "default" "LD_GN_RAW_C8=0"; do
if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
echo "== $cfg"; env $e timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^eager|^pipelined_list" | cut -c1-150
done; done
