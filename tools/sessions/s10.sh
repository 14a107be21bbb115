#!/bin/bash
# round 3, GPU session 10: GN row-wise reduce, BN backward with C8 output,
# pipelined bench
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bf16.py tests/test_gpu_e2e.py tests/test_gpu_graph.py tests/test_gpu_v2.py -q -m gpu -x > $O/s10_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s10_pytest.log
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_s10.json 2> $O/bench_s10.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s10.json').read().strip().split('\n')[-1])
print('fp32', round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'in-step teacher', round(d['config']['images_per_sec_teacher_in_step'],2), 'hits', d['config']['teacher_prefetch_hits'], 'roof', round(d['roofline']['frac'],3))
print('ldkl', round(d['roofline_ldkl']['frac'],3), round(d['roofline_ldkl']['frac_after_train_legs'],3))
print('bf16', round(d['bf16']['value'],1), round(d['bf16']['ms_per_step'],2), 'roof', round(d['roofline_bf16']['frac'],3), round(d['roofline_bf16']['conv_ms_per_step'],2), 'host', round(d['bf16']['host_enqueue_ms_per_step'],2))
print('graph', {k:round(v.get('value',0),1) for k,v in d['hipgraph_step'].items()})
PY
