#!/bin/bash
# round 5, session 14: the GPU tests that drive GradArena's bucket path after the
# in-order all-reduce change
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_graph_pg.py tests/test_gpu_defer.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -m gpu > gpurun_out/r05s14_tests.log 2>&1; echo rc=$?; tail -5 gpurun_out/r05s14_tests.log
LD_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-kernel-roofline > gpurun_out/r05s14_bench_fc.json 2> gpurun_out/r05s14_bench_fc.err; echo bench-fc rc=$?
python - <<'PY'
import json
e=json.loads(open('gpurun_out/r05s14_bench_fc.json').read().strip().splitlines()[-1])
print(e['value'],e['ms_per_step'],e.get('bf16',{}).get('value'))
PY
