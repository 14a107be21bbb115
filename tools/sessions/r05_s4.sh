#!/bin/bash
# round 5, session 4: fused bottleneck v2 (8-slot rings, hoisted epilogue loads, 128-channel
# chunks), deferred bias partials, thread-local capture mode, bench
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_defer.py tests/test_gpu_graph_pg.py tests/test_gpu_teacher_replay.py -q -x -m gpu > $O/r05s4_new.log 2>&1; echo new rc=$?; tail -12 $O/r05s4_new.log
timeout 300 python tools/bench_fused_block.py $O/r05_fused_block_v2.json > $O/r05s4_fused.log 2>&1; echo fused rc=$?; tail -1 $O/r05s4_fused.log
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_graph.py tests/test_gpu_rccl.py tests/test_gpu_fan.py -q -m gpu > $O/r05s4_suites.log 2>&1; echo suites rc=$?; tail -6 $O/r05s4_suites.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r05s4_bench.json 2> $O/r05s4_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s4_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'],'sync-median',d['config'].get('ms_per_step_synchronised_median'))
print('roofline',d['roofline']['frac'],d['roofline']['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in d['roofline']['by_kind'].items()})
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['host_enqueue_ms_per_step'],d['roofline_bf16']['frac'],d['roofline_bf16']['conv_ms_per_step'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1), v.get('error')) for m,v in d['hipgraph_step'].items()})
PY
