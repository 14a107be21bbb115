#!/bin/bash
# round 4, session 4: shipped wgrad table -- full GPU suite, gradient-element diagnostics, serial rocprof pass, full bench
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python tools/wgrad_sweep.py --only l2_conv1_first,head_tower --out $O/r04s4_sweep2.json > $O/r04s4_sweep2.log 2>&1; grep -v amdgpu.ids $O/r04s4_sweep2.log | tail -4
python - <<'PY'
import json
r=[x for x in json.load(open('gpurun_out/r04s4_sweep2.json')) if x['shape']=='l2_conv1_first' and 'us' in x]
r.sort(key=lambda x:x['us'])
for x in r[:6]: print(x['cand'],x['us'],x['tflops'],x['err'])
PY
timeout 1200 python -m pytest tests -q -m gpu -x --durations=6 > $O/r04s4_pytest.log 2>&1; echo pytest rc=$?; tail -12 $O/r04s4_pytest.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r04s4_serial -o step -- python /root/repo/tools/profile_step.py --mode fp32 --serial --steps 8 --warmup 2 > /root/repo/$O/r04s4_serial.log 2>&1); echo rocprof rc=$?; tail -2 $O/r04s4_serial.log
f=$(find $O/r04s4_serial -name '*kernel_stats.csv' | head -1); cp "$f" $O/r04s4_rocprof_kernel_stats_fp32_serial.csv; python tools/conv_frac_from_stats.py $O/r04s4_rocprof_kernel_stats_fp32_serial.csv --steps 10 | head -8
rm -rf $O/r04s4_serial
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04s4_bench.json 2> $O/r04s4_bench.err; echo bench rc=$?; tail -3 $O/r04s4_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s4_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'])
print('roofline',d['roofline']['frac'],d['roofline']['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in d['roofline']['by_kind'].items()})
print('bf16',d.get('bf16',{}).get('value'),d.get('roofline_bf16',{}).get('frac'))
print('graph',d.get('hipgraph_step'))
print('cpu',d.get('cpu_baseline'))
print('ldkl',d['roofline_ldkl']['frac'],d['roofline_ldkl'].get('frac_after_train_legs'))
PY
