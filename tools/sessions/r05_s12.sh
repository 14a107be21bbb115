#!/bin/bash
# round 5, session 12: the bench line again with this round's PMC constants and the
# fused bottleneck inside the bf16 roofline leg; the tests that touch KernelProfile
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_final.json 2> $O/r05f_bench.err; echo bench rc=$?
timeout 200 python tools/profile_step.py --mode bf16 --steps 6 --warmup 2 --layers $O/r05_layers_bf16.csv > $O/r05f_layers_bf16.log 2>&1; echo layers-bf16 rc=$?
timeout 600 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_layers.py tests/test_gpu_teacher_replay.py -q -m gpu > $O/r05s12_tests.log 2>&1; echo tests rc=$?; tail -3 $O/r05s12_tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_final.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'],'sync-median',d['config'].get('ms_per_step_synchronised_median'))
r=d['roofline']
print('roofline',r['frac'],r['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in r['by_kind'].items()})
print({k:v for k,v in r.items() if 'over' in k or 'fused' in k or k=='traffic'})
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['host_enqueue_ms_per_step'],d['roofline_bf16']['frac'],d['roofline_bf16']['conv_ms_per_step'],d['roofline_bf16']['gflop_per_step'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for m,v in d['hipgraph_step'].items()})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['all_totals_s'])
PY
