#!/bin/bash
# round 3, GPU session 7: full GPU suite on the fused conv+BN forward, C8-only
# frozen student stages, streaming stem, vector max-pool; bench; bucket timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x --durations=10 > $O/s7_pytest.log 2>&1; echo pytest rc=$?; tail -4 $O/s7_pytest.log
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_s7.json 2> $O/bench_s7.err; echo bench rc=$?
timeout 200 python tools/profile_step.py --mode fp32 --steps 3 --buckets $O/bucket_timeline_fp32.txt > $O/s7_buckets.log 2>&1; tail -8 $O/bucket_timeline_fp32.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s7.json').read().strip().split('\n')[-1])
print('fp32', round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'roof', round(d['roofline']['frac'],3), {k:(round(v['tflops'],1),round(v['ms_per_step'],2)) for k,v in d['roofline']['by_kind'].items()})
print('ldkl', round(d['roofline_ldkl']['frac'],3), round(d['roofline_ldkl']['frac_after_train_legs'],3))
print('bf16', round(d['bf16']['value'],1), round(d['bf16']['ms_per_step'],2), 'roof', round(d['roofline_bf16']['frac'],3), round(d['roofline_bf16']['conv_ms_per_step'],2))
print('graph', {k:round(v.get('value',0),1) for k,v in d['hipgraph_step'].items()})
PY
