#!/bin/bash
# round 5, final session: the judged numbers first (bench lines, rocprofv3 kernel
# statistics, launches per step, per-layer tables, PMC traffic), then smoke() and
# the full GPU suite on the same build
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
R=/root/repo
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_final.json 2> $O/r05f_bench.err; echo bench rc=$?
LD_FORCE_COLLECTIVES=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $O/r05_bench_torchrun_1rank_forced_collectives_final.json 2> $O/r05f_bench_fc.err; echo bench-fc rc=$?
for c in 4 5; do
timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r05_bench_final_config$c.json 2> $O/r05f_bench_c$c.err; echo bench-c$c rc=$?
done
# serialised fp32 step: kernel statistics -> roofline.frac recomputed from the profile
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r05f_serial -o step -- python $R/tools/profile_step.py --mode fp32 --serial --steps 8 --warmup 2 > $R/$O/r05f_serial.log 2>&1)
f=$(find $O/r05f_serial -name '*kernel_stats.csv' | head -1); cp "$f" $O/r05_rocprof_kernel_stats_fp32_serial.csv
python tools/conv_frac_from_stats.py $O/r05_rocprof_kernel_stats_fp32_serial.csv --steps 10 > $O/r05_conv_frac_from_stats.txt 2>&1; tail -3 $O/r05_conv_frac_from_stats.txt
rm -rf $O/r05f_serial
# overlapped step, both modes: kernel statistics + launches per step from the trace
for m in fp32 bf16; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r05f_prof_$m -o step -- python $R/tools/profile_step.py --mode $m --steps 10 --warmup 3 --pipeline > $R/$O/r05f_prof_$m.log 2>&1)
f=$(find $O/r05f_prof_$m -name '*kernel_stats.csv' | head -1); cp "$f" $O/r05_rocprof_kernel_stats_$m.csv
t=$(find $O/r05f_prof_$m -name '*kernel_trace.csv' | head -1)
python tools/launches_per_step.py "$t" --steps 5 > $O/r05_launches_per_step_$m.txt 2>&1; head -4 $O/r05_launches_per_step_$m.txt
rm -rf $O/r05f_prof_$m
timeout 200 python tools/profile_step.py --mode $m --steps 6 --warmup 2 --layers $O/r05_layers_$m.csv > $O/r05f_layers_$m.log 2>&1; echo layers-$m rc=$?
done
# fabric-side traffic of the conv kernels of the serialised fp32 step, by kernel
PMC_BY_KERNEL=1 timeout 500 bash tools/pmc_traffic.sh r05_conv_step_fp32_by_kernel conv_ -- python $R/tools/profile_step.py --mode fp32 --serial --steps 4 --warmup 2 > $O/r05f_pmc.log 2>&1; echo pmc rc=$?
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05f_smoke.log 2>&1; echo smoke rc=$?; tail -2 $O/r05f_smoke.log
timeout 1200 python -m pytest tests -q -m gpu > $O/r05_pytest_gpu_final.txt 2>&1; echo pytest rc=$?; tail -4 $O/r05_pytest_gpu_final.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_final.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'in-step',d['config']['images_per_sec_teacher_in_step'],'enq',d['config']['host_enqueue_ms_per_step'],'sync-median',d['config'].get('ms_per_step_synchronised_median'))
r=d['roofline']
print('roofline',r['frac'],r['conv_ms_per_step'],{k:(round(v['ms_per_step'],2),round(v['tflops'],1)) for k,v in r['by_kind'].items()},r.get('traffic_over_algorithmic'),r.get('fetch_over_algorithmic_reads'),r.get('write_over_algorithmic_writes'))
print('bf16',d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['host_enqueue_ms_per_step'],d['roofline_bf16']['frac'])
print('graph',{m:(round(v.get('value',0),1),round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for m,v in d['hipgraph_step'].items()})
print('cpu',d['cpu_baseline'])
for n in ('r05_bench_torchrun_1rank_forced_collectives_final','r05_bench_final_config4','r05_bench_final_config5'):
    try:
        e=json.loads(open(f'gpurun_out/{n}.json').read().strip().splitlines()[-1])
        print(n,e['value'],e['ms_per_step'],e.get('bf16',{}).get('value'))
    except Exception as ex: print(n,'failed',ex)
PY
