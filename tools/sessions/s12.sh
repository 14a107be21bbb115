#!/bin/bash
# round 3, GPU session 12: pipelined hipGraphs
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_graph.py -q -m gpu -x > $O/s12_pytest.log 2>&1; echo pytest rc=$?; tail -15 $O/s12_pytest.log | cut -c1-200
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s12.json 2> $O/bench_s12.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s12.json').read().strip().split('\n')[-1])
print('fp32', round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'in-step teacher', round(d['config']['images_per_sec_teacher_in_step'],2), 'roof', round(d['roofline']['frac'],3), 'host', round(d['config']['host_enqueue_ms_per_step'],2))
print('bf16', round(d['bf16']['value'],1), round(d['bf16']['ms_per_step'],2), 'host', round(d['bf16']['host_enqueue_ms_per_step'],2))
print('graph', json.dumps(d['hipgraph_step'])[:900])
PY
