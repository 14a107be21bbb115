#!/bin/bash
# round 6, GPU session 14: A/B with enough repetitions to see through the bimodal
# run-to-run behaviour (some processes land ~0.7 ms slower for their whole run)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
: > $O/s14_runs.txt
for rep in 1 2 3 4 5 6; do
for cfg in "default" "LD_BN_LEAN=0"; do
if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
r=$(env $e timeout 200 python tools/profile_step.py --mode bf16 --steps 60 --warmup 10 --pipeline 2>/dev/null | grep img/s | sed 's/.*: \([0-9.]*\) ms.*/\1/')
echo "$cfg $r" | tee -a $O/s14_runs.txt
done; done
python - <<'PY'
import collections, statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/s14_runs.txt'):
    *k,v=l.split(); d[' '.join(k)].append(float(v))
for k,v in d.items():
    print(k, 'min %.2f median %.2f max %.2f'%(min(v),statistics.median(v),max(v)), sorted(v))
PY
