#!/bin/bash
# round 6, GPU session 23: zero-padded dY for the 68-channel data gradient (gfl_reg):
# tests, band, A/B through the step list
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -m gpu -x > $O/s23_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s23_pytest.log
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s -k "golden" > $O/s23_band.log 2>&1; grep -E "grad-norm relative|passed|failed" $O/s23_band.log
for rep in 1 2 3; do
for cfg in "default" "LD_PAD_DGRAD=0"; do
if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
echo "== $cfg"; env $e timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^pipelined_list" | cut -c1-120
done; done
