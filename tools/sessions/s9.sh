#!/bin/bash
# round 3, GPU session 9: vector 1x1 conv kernel -- tests, re-tune, layer table,
# bench
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_layers.py -q -m gpu -x > $O/s9_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s9_pytest.log
timeout 900 python tools/tune_conv.py --fresh-family 0 --modes fp32 --student 50 --out $O/tune_r03b.txt > $O/s9_tune.log 2>&1; echo tune rc=$?; tail -1 $O/s9_tune.log
grep -c "x2 cap" $O/s9_tune.log; grep "x2 cap" $O/s9_tune.log | head -40
export LD_CONV_TUNE_FILE=$R/$O/tune_r03b.txt
timeout 300 python tools/profile_step.py --mode fp32 --steps 4 --layers $O/layers_fp32_r03b.csv > $O/s9_layers.log 2>&1; tail -2 $O/s9_layers.log
timeout 200 python tools/profile_step.py --mode fp32 --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
timeout 300 python bench.py --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/bench_s9.json 2> $O/bench_s9.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s9.json').read().strip().split('\n')[-1])
print('fp32', round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'roof', round(d['roofline']['frac'],3), {k:(round(v['tflops'],1),round(v['ms_per_step'],2)) for k,v in d['roofline']['by_kind'].items()})
PY
