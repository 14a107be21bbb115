#!/bin/bash
# round 4, session 3: tap3 wgrad kernel -- tests + sweep + A/B bench
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_layers.py -q -x -k "wgrad" > $O/r04s3_pytest.log 2>&1; echo pytest rc=$?; tail -5 $O/r04s3_pytest.log
timeout 900 python tools/wgrad_sweep.py --out $O/r04s3_wgrad_sweep.json --table $O/r04s3_wgrad_table.txt > $O/r04s3_sweep.log 2>&1; echo sweep rc=$?; grep -v amdgpu.ids $O/r04s3_sweep.log | tail -40
LD_CONV_WGRAD_CFG=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s3_bench_old.json 2> $O/r04s3_bench_old.err; echo bench_old rc=$?
LD_CONV_TUNE_FILE=$O/r04s3_wgrad_table.txt timeout 400 python bench.py --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s3_bench_new.json 2> $O/r04s3_bench_new.err; echo bench_new rc=$?
python - <<'PY'
import json
for n in ('old','new'):
    try:
        d=json.loads(open(f'gpurun_out/r04s3_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('by_kind'))
    except Exception as e:
        print(n, 'ERR', e)
PY
