#!/bin/bash
# round 3, GPU session 6: re-tune the fp32 streaming shapes with the residency
# cap as a tuned field, per-layer table + step time with the new table, LD-KL
# PMC traffic, config-4 test
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python tools/tune_conv.py --fresh-family 0 --modes fp32 --student 50 --out $O/tune_r03_a.txt > $O/s6_tune_a.log 2>&1; echo tune50 rc=$?; tail -2 $O/s6_tune_a.log
LD_CONV_TUNE_FILE=$R/$O/tune_r03_a.txt timeout 600 python tools/tune_conv.py --modes fp32 --student 101 --out $O/tune_r03.txt > $O/s6_tune_b.log 2>&1; echo tune101 rc=$?; tail -2 $O/s6_tune_b.log
grep -c "cap [234]" $O/s6_tune_a.log
export LD_CONV_TUNE_FILE=$R/$O/tune_r03.txt
timeout 300 python tools/profile_step.py --mode fp32 --steps 4 --layers $O/layers_fp32_r03.csv > $O/s6_layers.log 2>&1; tail -3 $O/s6_layers.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/bench_s6.json 2> $O/bench_s6.err; echo bench rc=$?
timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --no-bf16 --no-graph > $O/bench_s6_config4.json 2> $O/bench_s6_config4.err; echo bench4 rc=$?
timeout 200 python -m pytest tests/test_gpu_config4.py tests/test_gpu_layers.py -q -m gpu -x > $O/s6_pytest.log 2>&1; echo pytest rc=$?; tail -2 $O/s6_pytest.log
unset LD_CONV_TUNE_FILE
timeout 300 bash tools/pmc_traffic.sh regdense_r3 loss_reg_lean -- python $R/tools/one_regdense.py > $O/s6_pmc.log 2>&1; cat $O/pmc_traffic_regdense_r3.txt
python - <<'PY'
import json
for f in ('bench_s6','bench_s6_config4'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().split('\n')[-1])
        print(f, round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'roof', round(d.get('roofline',{}).get('frac',0),3), {k:round(v['tflops'],1) for k,v in d['roofline']['by_kind'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
