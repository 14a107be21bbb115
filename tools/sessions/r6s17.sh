#!/bin/bash
# round 6, GPU session 17: tower y-skip tests, bf16/graph/e2e suites, A/B, step-list timing
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bf16.py tests/test_gpu_graph.py tests/test_gpu_e2e.py tests/test_gpu_v2.py -q -m gpu -x > $O/s17_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s17_pytest.log
: > $O/s17_runs.txt
for rep in 1 2 3 4 5; do
for cfg in "default" "LD_TRUNK_C8=0"; do
if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
r=$(env $e timeout 200 python tools/profile_step.py --mode bf16 --steps 60 --warmup 10 --pipeline 2>/dev/null | grep img/s | sed 's/.*: \([0-9.]*\) ms.*/\1/')
echo "$cfg $r" | tee -a $O/s17_runs.txt
done; done
python - <<'PY'
import collections, statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/s17_runs.txt'):
    *k,v=l.split(); d[' '.join(k)].append(float(v))
for k,v in d.items():
    print(k, 'min %.2f median %.2f max %.2f'%(min(v),statistics.median(v),max(v)), sorted(v))
PY
timeout 600 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^eager|^pipelined|^single" | cut -c1-220
