#!/bin/bash
# round 3 final validation (second build): full GPU suite + smoke, then the bench lines
# (driver flags), config 4 / 5, rocprof kernel stats of both modes
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1000 python -m pytest tests -q -m gpu --durations=12 > $O/final2_pytest.log 2>&1; echo pytest rc=$?; grep -E "passed|failed" $O/final2_pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/final2_smoke.log 2>&1; echo smoke rc=$?; tail -1 $O/final2_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final2.json 2> $O/bench_final2.err; echo bench rc=$?
timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --no-bf16 --no-graph > $O/bench_final2_config4.json 2> $O/bench_final2_config4.err; echo bench4 rc=$?
timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --no-bf16 --no-graph > $O/bench_final2_config5.json 2> $O/bench_final2_config5.err; echo bench5 rc=$?
for m in fp32 bf16; do
  rm -rf /tmp/rp_$m
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/rp_$m -o st -- python $R/tools/profile_step.py --mode $m --steps 10 --warmup 3 --pipeline > $R/$O/final2_rocprof_$m.log 2>&1 )
  DB=$(find /tmp/rp_$m -name '*.db' | head -1)
  python tools/rocpd_stats.py $DB $O/final2_rocprof_kernel_stats_$m.csv
done
python - <<'PY'
import json
for f in ('bench_final2','bench_final2_config4','bench_final2_config5'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().split('\n')[-1])
        print(f, round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms in-step', round(d['config']['images_per_sec_teacher_in_step'] or 0,2), 'roof', round(d.get('roofline',{}).get('frac',0),3), 'ldkl', round(d.get('roofline_ldkl',{}).get('frac',0),3), round(d.get('roofline_ldkl',{}).get('frac_after_train_legs',0),3), 'bf16', round(d.get('bf16',{}).get('value',0),1))
        if 'hipgraph_step' in d: print('   graph', {k:(round(v.get('value',0),1), round(v.get('teacher_one_step_ahead',{}).get('value',0),1)) for k,v in d['hipgraph_step'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
