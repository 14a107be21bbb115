#!/bin/bash
# round 4, session 19: final evidence on the final build
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $O/r04s19_pytest.log 2>&1; echo pytest rc=$?; grep -E "passed|failed" $O/r04s19_pytest.log | tail -1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04s19_smoke.log 2>&1; echo smoke rc=$?; tail -1 $O/r04s19_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04s19_bench.json 2> $O/r04s19_bench.err; echo bench rc=$?
timeout 400 python bench.py --config 4 --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s19_bench_config4.json 2>> $O/r04s19_bench.err; echo bench4 rc=$?
timeout 400 python bench.py --config 5 --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s19_bench_config5.json 2>> $O/r04s19_bench.err; echo bench5 rc=$?
LD_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 20 --warmup 5 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s19_torchrun.json 2> $O/r04s19_torchrun.err; echo torchrun rc=$?
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r04s19_serial -o step -- python /root/repo/tools/profile_step.py --mode fp32 --serial --steps 8 --warmup 2 > /root/repo/$O/r04s19_serial.log 2>&1)
f=$(find $O/r04s19_serial -name '*kernel_stats.csv' | head -1); cp "$f" $O/r04s19_rocprof_kernel_stats_fp32_serial.csv; rm -rf $O/r04s19_serial; python tools/conv_frac_from_stats.py $O/r04s19_rocprof_kernel_stats_fp32_serial.csv --steps 10 | head -2
python - <<'PY'
import json
def L(f): return json.loads(open(f).read().strip().splitlines()[-1])
d=L('gpurun_out/r04s19_bench.json')
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'])
print('roofline',d['roofline']['frac'],d['roofline']['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in d['roofline']['by_kind'].items()})
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['roofline_bf16']['frac'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for m,v in d['hipgraph_step'].items()})
print('cpu',d['cpu_baseline']['kind'],d['cpu_baseline']['value'],d['cpu_baseline']['stages_s'])
print('ldkl',d['roofline_ldkl']['frac'],d['roofline_ldkl'].get('frac_after_train_legs'))
for c in (4,5):
    e=L(f'gpurun_out/r04s19_bench_config{c}.json'); print('config',c,e['value'],e['ms_per_step'],e['roofline']['frac'])
t=L('gpurun_out/r04s19_torchrun.json'); print('torchrun forced collectives',t['value'],t['ms_per_step'],t['config']['images_per_sec_teacher_in_step'])
PY
