#!/bin/bash
# round 5, session 7: confirm the scratch-buffer fix (stress), changed tests, fpn bias dy
# diagnostic, bench
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 400 python tools/stress_pipelined.py 24 bf16 2>&1 | grep -v amdgpu.ids | tail -6 > $O/r05_stress_pipelined_after_fix.txt; cat $O/r05_stress_pipelined_after_fix.txt
timeout 900 python -m pytest tests/test_gpu_graph_pg.py tests/test_gpu_fan.py tests/test_gpu_teacher_replay.py tests/test_gpu_graph.py tests/test_gpu_bf16.py tests/test_gpu_e2e.py tests/test_gpu_rccl.py -q -m gpu > $O/r05s7_suites.log 2>&1; echo suites rc=$?; tail -8 $O/r05s7_suites.log
timeout 300 python tools/fpn_bias_dy.py > $O/r05_fpn_bias_dy.txt 2>$O/r05s7_fpn.err; echo fpn rc=$?; cat $O/r05_fpn_bias_dy.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r05s7_bench.json 2> $O/r05s7_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s7_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'],'sync-median',d['config'].get('ms_per_step_synchronised_median'))
print('roofline',d['roofline']['frac'],d['roofline']['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in d['roofline']['by_kind'].items()})
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['host_enqueue_ms_per_step'],d['roofline_bf16']['frac'],d['roofline_bf16']['conv_ms_per_step'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1), v.get('error')) for m,v in d['hipgraph_step'].items()})
PY
