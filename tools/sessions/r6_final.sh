#!/bin/bash
# round 6, final session: the judged numbers (bench lines, rocprofv3 kernel statistics
# of the serialised and the overlapped step, launches per step, per-layer tables),
# smoke() and the full GPU suite on the same build
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_final.json 2> $O/r06f_bench.err; echo bench rc=$?
LD_FORCE_COLLECTIVES=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $O/r06_bench_torchrun_1rank_forced_collectives_final.json 2> $O/r06f_bench_fc.err; echo bench-fc rc=$?
for c in 4 5; do
timeout 400 python bench.py --config $c --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r06_bench_final_config$c.json 2> $O/r06f_bench_c$c.err; echo bench-c$c rc=$?
done
# serialised fp32 step: kernel statistics -> roofline.frac recomputed from the profile
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06f_serial -o step -- python $R/tools/profile_step.py --mode fp32 --serial --steps 8 --warmup 2 > $R/$O/r06f_serial.log 2>&1)
f=$(find $O/r06f_serial -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06_rocprof_kernel_stats_fp32_serial.csv
python tools/conv_frac_from_stats.py $O/r06_rocprof_kernel_stats_fp32_serial.csv --steps 10 > $O/r06_conv_frac_from_stats.txt 2>&1; tail -3 $O/r06_conv_frac_from_stats.txt
rm -rf $O/r06f_serial
# the same for bf16 (serialised): per-kernel durations of the bf16 conv family
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06f_serialb -o step -- python $R/tools/profile_step.py --mode bf16 --serial --steps 8 --warmup 2 > $R/$O/r06f_serialb.log 2>&1)
f=$(find $O/r06f_serialb -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06_rocprof_kernel_stats_bf16_serial.csv; rm -rf $O/r06f_serialb
# overlapped step, both modes: kernel statistics + launches per step + queue occupancy
for m in fp32 bf16; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06f_prof_$m -o step -- python $R/tools/profile_step.py --mode $m --steps 10 --warmup 3 --pipeline > $R/$O/r06f_prof_$m.log 2>&1)
f=$(find $O/r06f_prof_$m -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06_rocprof_kernel_stats_$m.csv
t=$(find $O/r06f_prof_$m -name '*kernel_trace.csv' | head -1)
python tools/launches_per_step.py "$t" --steps 5 > $O/r06_launches_per_step_$m.txt 2>&1; head -3 $O/r06_launches_per_step_$m.txt
python tools/queue_busy.py "$t" --steps 5 > $O/r06_queue_busy_$m.txt 2>&1
rm -rf $O/r06f_prof_$m
timeout 200 python tools/profile_step.py --mode $m --steps 6 --warmup 2 --layers $O/r06_layers_$m.csv > $O/r06f_layers_$m.log 2>&1; echo layers-$m rc=$?
done
timeout 300 python tools/bench_t256.py head $O/r06_t256_head.json 2>&1 | grep shape
timeout 300 python tools/bench_step_list.py bf16 20 $O/r06_step_list_bf16.json 2>&1 | grep -E "eager|pipelined_list"
timeout 300 python tools/bench_step_list.py fp32 20 $O/r06_step_list_fp32.json 2>&1 | grep -E "eager|pipelined_list"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r06_smoke_final.txt 2>&1; echo smoke rc=$?; tail -2 $O/r06_smoke_final.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/r06_pytest_gpu_final.txt 2>&1; echo pytest rc=$?; tail -4 $O/r06_pytest_gpu_final.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_final.json').read().strip().splitlines()[-1])
c=d['config']
print('value',d['value'],'ms',d['ms_per_step'],'bracket',d['images_per_sec_k_step_bracket'],'in-step',c['images_per_sec_teacher_in_step'],'enq',c['host_enqueue_ms_per_step'])
r=d['roofline']
print('roofline',r['frac'],r['achieved'],r['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in r['by_kind'].items()},r.get('traffic_over_algorithmic'),r.get('fetch_over_algorithmic_reads'),r.get('write_over_algorithmic_writes'))
print('ldkl',d['roofline_ldkl']['frac'],d['roofline_ldkl'].get('frac_after_train_legs'))
b=d['bf16']; rb=d['roofline_bf16']
print('bf16',b['value'],b['ms_per_step'],b['images_per_sec_k_step_bracket'],b['host_enqueue_ms_per_step'],'roof',rb['frac'],rb['achieved'],rb['traffic_over_algorithmic'],rb['launches_per_step'])
print('step_list',{m:(round(v.get('value',0),1),round(v.get('host_ms_one_replay_idle_queue',0),2)) if 'value' in v else v for m,v in d['step_list'].items()})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['cores'],d['cpu_baseline']['kind'])
for n in ('r06_bench_torchrun_1rank_forced_collectives_final','r06_bench_final_config4','r06_bench_final_config5'):
    try:
        e=json.loads(open(f'gpurun_out/{n}.json').read().strip().splitlines()[-1])
        print(n,e['value'],e['ms_per_step'],e['images_per_sec_k_step_bracket'],e.get('bf16',{}).get('value'),e['roofline']['frac'],e['config'].get('parity_note'))
    except Exception as ex: print(n,'failed',ex)
PY
