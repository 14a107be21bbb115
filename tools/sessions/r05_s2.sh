#!/bin/bash
# round 5, session 2: the new fused-accumulation / deferred-finalisation / process-group
# graph tests, the suites they touch, where the copyBuffer launches come from, a bench
# line and per-kernel launch counts of both modes
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_fan.py tests/test_gpu_defer.py tests/test_gpu_graph_pg.py -q -x -m gpu > $O/r05s2_new.log 2>&1; echo new rc=$?; tail -25 $O/r05s2_new.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_graph.py tests/test_gpu_rccl.py tests/test_gpu_bf16.py tests/test_gpu_v2.py -q -m gpu > $O/r05s2_suites.log 2>&1; echo suites rc=$?; tail -12 $O/r05s2_suites.log
timeout 300 python tools/find_memcpy.py fp32 > $O/r05_find_memcpy_fp32.txt 2>&1; echo fm rc=$?; tail -45 $O/r05_find_memcpy_fp32.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r05s2_bench.json 2> $O/r05s2_bench.err; echo bench rc=$?
for m in fp32 bf16; do
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r05s2_prof_$m -o step -- python /root/repo/tools/profile_step.py --mode $m --steps 10 --warmup 3 --pipeline > /root/repo/$O/r05s2_prof_$m.log 2>&1)
f=$(find $O/r05s2_prof_$m -name '*kernel_stats.csv' | head -1); cp "$f" $O/r05s2_rocprof_kernel_stats_$m.csv; rm -rf $O/r05s2_prof_$m
done
python - <<'PY'
import json,csv
d=json.loads(open('gpurun_out/r05s2_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'],'sync-median',d['config'].get('ms_per_step_synchronised_median'))
print('roofline',d['roofline']['frac'],d['roofline']['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in d['roofline']['by_kind'].items()})
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['host_enqueue_ms_per_step'],d['roofline_bf16']['frac'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for m,v in d['hipgraph_step'].items()})
for m in ('fp32','bf16'):
    rows=list(csv.DictReader(open(f'gpurun_out/r05s2_rocprof_kernel_stats_{m}.csv')))
    calls=sum(int(r['Calls']) for r in rows); tot=sum(int(r['TotalDurationNs']) for r in rows)
    print(m,'launches total',calls,'kernel ms total',tot/1e6)
    for r in rows[:14]: print('   ',r['Calls'],r['TotalDurationNs'],r['Name'][:90])
PY
