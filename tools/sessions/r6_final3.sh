#!/bin/bash
# round 6, last session: the judged numbers again
# on the build with the re-tuned C8 shape table (bench line, bf16 profiles, step lists,
# smoke, whole GPU suite)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_final.json 2> $O/r06f_bench.err; echo bench rc=$?
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06f_serialb -o step -- python $R/tools/profile_step.py --mode bf16 --serial --steps 8 --warmup 2 > $R/$O/r06f_serialb.log 2>&1)
f=$(find $O/r06f_serialb -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06_rocprof_kernel_stats_bf16_serial.csv; rm -rf $O/r06f_serialb
m=bf16
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06f_prof_$m -o step -- python $R/tools/profile_step.py --mode $m --steps 10 --warmup 3 --pipeline > $R/$O/r06f_prof_$m.log 2>&1)
f=$(find $O/r06f_prof_$m -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06_rocprof_kernel_stats_$m.csv
t=$(find $O/r06f_prof_$m -name '*kernel_trace.csv' | head -1)
python tools/launches_per_step.py "$t" --steps 5 > $O/r06_launches_per_step_$m.txt 2>&1; head -3 $O/r06_launches_per_step_$m.txt
python tools/queue_busy.py "$t" --steps 5 > $O/r06_queue_busy_$m.txt 2>&1
rm -rf $O/r06f_prof_$m
timeout 200 python tools/profile_step.py --mode $m --steps 6 --warmup 2 --layers $O/r06_layers_$m.csv > $O/r06f_layers_$m.log 2>&1; echo layers-$m rc=$?
timeout 300 python tools/bench_step_list.py bf16 20 $O/r06_step_list_bf16.json 2>&1 | grep -E "eager|pipelined_list"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r06_smoke_final.txt 2>&1; echo smoke rc=$?; tail -2 $O/r06_smoke_final.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/r06_pytest_gpu_final.txt 2>&1; echo pytest rc=$?; tail -4 $O/r06_pytest_gpu_final.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_final.json').read().strip().splitlines()[-1])
c=d['config']
print('value',d['value'],'ms',d['ms_per_step'],'bracket',d['images_per_sec_k_step_bracket'],'in-step',c['images_per_sec_teacher_in_step'],'enq',c['host_enqueue_ms_per_step'])
r=d['roofline']
print('roofline',r['frac'],r['achieved'],r['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in r['by_kind'].items()},r.get('traffic_over_algorithmic'))
print('ldkl',d['roofline_ldkl']['frac'],d['roofline_ldkl'].get('frac_after_train_legs'))
b=d['bf16']; rb=d['roofline_bf16']
print('bf16',b['value'],b['ms_per_step'],b['images_per_sec_k_step_bracket'],b['host_enqueue_ms_per_step'],b['host_bound'],'roof',rb['frac'],rb['achieved'],rb['conv_ms_per_step'],rb['traffic_over_algorithmic'],rb['launches_per_step'])
print('step_list',{m:(round(v.get('value',0),1),round(v.get('host_ms_one_replay_idle_queue',0),2)) if 'value' in v else v for m,v in d['step_list'].items()})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['cores'],d['cpu_baseline']['kind'])
PY
