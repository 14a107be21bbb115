#!/bin/bash
# round 3, GPU session 11: weight gradients on a side stream
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_graph.py tests/test_gpu_rccl.py tests/test_gpu_v2.py tests/test_gpu_atss.py -q -m gpu -x > $O/s11_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s11_pytest.log
for m in fp32 bf16; do
  LD_WGRAD_STREAM=0 timeout 200 python tools/profile_step.py --mode $m --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
  LD_WGRAD_STREAM=1 timeout 200 python tools/profile_step.py --mode $m --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
done | tee $O/s11_wgrad_stream.txt
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s11.json 2> $O/bench_s11.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s11.json').read().strip().split('\n')[-1])
print('fp32', round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'in-step teacher', round(d['config']['images_per_sec_teacher_in_step'],2), 'roof', round(d['roofline']['frac'],3), 'host', round(d['config']['host_enqueue_ms_per_step'],2))
print('bf16', round(d['bf16']['value'],1), round(d['bf16']['ms_per_step'],2), 'host', round(d['bf16']['host_enqueue_ms_per_step'],2))
print('graph', {k:round(v.get('value',0),1) for k,v in d['hipgraph_step'].items()})
PY
