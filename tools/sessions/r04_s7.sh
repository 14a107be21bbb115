#!/bin/bash
# round 4, session 7: the round's evidence run on the final build
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $O/r04s7_pytest.log 2>&1; echo pytest rc=$?; tail -4 $O/r04s7_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04s7_smoke.log 2>&1; echo smoke rc=$?; tail -1 $O/r04s7_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04s7_bench.json 2> $O/r04s7_bench.err; echo bench rc=$?
timeout 400 python bench.py --config 4 --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s7_bench_config4.json 2>> $O/r04s7_bench.err; echo bench4 rc=$?
timeout 400 python bench.py --config 5 --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s7_bench_config5.json 2>> $O/r04s7_bench.err; echo bench5 rc=$?
timeout 300 python tools/profile_step.py --mode fp32 --steps 15 --layers $O/r04s7_layers_fp32.csv > $O/r04s7_layers_fp32.log 2>&1; tail -1 $O/r04s7_layers_fp32.log
timeout 300 python tools/profile_step.py --mode bf16 --steps 15 --layers $O/r04s7_layers_bf16.csv > $O/r04s7_layers_bf16.log 2>&1; tail -1 $O/r04s7_layers_bf16.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r04s7_serial -o step -- python /root/repo/tools/profile_step.py --mode fp32 --serial --steps 8 --warmup 2 > /root/repo/$O/r04s7_serial.log 2>&1)
f=$(find $O/r04s7_serial -name '*kernel_stats.csv' | head -1); cp "$f" $O/r04s7_rocprof_kernel_stats_fp32_serial.csv; rm -rf $O/r04s7_serial; python tools/conv_frac_from_stats.py $O/r04s7_rocprof_kernel_stats_fp32_serial.csv --steps 10 | head -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r04s7_ovl -o step -- python /root/repo/tools/profile_step.py --mode fp32 --pipeline --steps 10 --warmup 3 > /root/repo/$O/r04s7_ovl.log 2>&1)
f=$(find $O/r04s7_ovl -name '*kernel_stats.csv' | head -1); cp "$f" $O/r04s7_rocprof_kernel_stats_fp32.csv; rm -rf $O/r04s7_ovl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r04s7_ovlb -o step -- python /root/repo/tools/profile_step.py --mode bf16 --pipeline --steps 10 --warmup 3 > /root/repo/$O/r04s7_ovlb.log 2>&1)
f=$(find $O/r04s7_ovlb -name '*kernel_stats.csv' | head -1); cp "$f" $O/r04s7_rocprof_kernel_stats_bf16.csv; rm -rf $O/r04s7_ovlb
PMC_BY_KERNEL=1 timeout 600 bash tools/pmc_traffic.sh r04_conv_step_fp32_by_kernel "conv_" -- python /root/repo/tools/profile_step.py --mode fp32 --serial --steps 4 --warmup 2 > $O/r04s7_pmc.log 2>&1
head -30 $O/pmc_traffic_r04_conv_step_fp32_by_kernel.txt | cut -c1-230
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s7_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'])
print('roofline',d['roofline']['frac'],d['roofline']['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in d['roofline']['by_kind'].items()})
print('bf16',d.get('bf16',{}).get('value'),d.get('roofline_bf16',{}).get('frac'))
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for m,v in d.get('hipgraph_step',{}).items()})
print('cpu',d.get('cpu_baseline',{}).get('kind'),d.get('cpu_baseline',{}).get('value'))
print('ldkl',d['roofline_ldkl']['frac'],d['roofline_ldkl'].get('frac_after_train_legs'))
for c in (4,5):
    e=json.loads(open(f'gpurun_out/r04s7_bench_config{c}.json').read().strip().splitlines()[-1])
    print('config',c,e['value'],e['ms_per_step'],e['roofline']['frac'])
PY
