#!/bin/bash
# round 6, GPU session 4: new tests (half-step DDP semantics, tightened bf16 band,
# list launcher), bench.py with the new fields
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ddp_halves.py tests/test_gpu_graph.py "tests/test_gpu_bf16.py::test_bf16_train_step_vs_fp32_golden" -q -m gpu -x -s > $O/s4_pytest.log 2>&1; echo pytest rc=$?; grep -E "worst relative|grad-norm relative|passed|failed|Error" $O/s4_pytest.log | tail -12
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_s4.json 2> $O/bench_s4.err; echo bench rc=$?; tail -3 $O/bench_s4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s4.json').read().strip().split('\n')[-1])
c=d['config']
print('fp32 value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'bracket', round(d['images_per_sec_k_step_bracket'],2), 'host', round(c['host_enqueue_ms_per_step'],2), 'roof', round(d['roofline']['frac'],3))
b=d.get('bf16',{}); print('bf16', {k:(round(v,2) if isinstance(v,float) else v) for k,v in b.items() if k in ('value','ms_per_step','images_per_sec_k_step_bracket','host_enqueue_ms_per_step')})
print('roof bf16', round(d['roofline_bf16']['frac'],4), 'traffic ratio', d['roofline_bf16']['traffic_over_algorithmic'])
print('step_list', json.dumps(d.get('step_list'))[:900])
PY
