#!/bin/bash
# round 5, session 18: the bench line on the build with the bf16 weight gradients on
# the side stream (no CPU baseline: that leg is unchanged, r05_bench_final.json)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 280 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_final_bf16_wgrad_side.json 2> gpurun_out/r05s18_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_final_bf16_wgrad_side.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'enq',d['config']['host_enqueue_ms_per_step'])
print('roofline',d['roofline']['frac'])
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['host_enqueue_ms_per_step'],d['roofline_bf16']['frac'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for m,v in d['hipgraph_step'].items()})
PY
