#!/bin/bash
# round 6, GPU session 18: which of the placeholder-output changes costs time in the
# eager step: default / LD_GN_YSKIP=0 / LD_TRUNK_C8=0, alternating; host profile
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
: > $O/s18_runs.txt
for rep in 1 2 3 4; do
for cfg in "default" "LD_GN_YSKIP=0" "LD_TRUNK_C8=0"; do
if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
r=$(env $e timeout 200 python tools/profile_step.py --mode bf16 --steps 60 --warmup 10 --pipeline 2>/dev/null | grep img/s | sed 's/.*: \([0-9.]*\) ms.*/\1/')
echo "$cfg $r" | tee -a $O/s18_runs.txt
done; done
python - <<'PY'
import collections, statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/s18_runs.txt'):
    *k,v=l.split(); d[' '.join(k)].append(float(v))
for k,v in d.items():
    print(k, 'min %.2f median %.2f max %.2f'%(min(v),statistics.median(v),max(v)), sorted(v))
PY
for cfg in "default" "LD_GN_YSKIP=0"; do
if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
echo "== bench_step_list $cfg"; env $e timeout 600 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^eager|^pipelined_list" | cut -c1-200
done
timeout 300 python tools/host_profile.py bf16 20 > $O/s18_host_profile.txt 2>&1; head -60 $O/s18_host_profile.txt
