#!/bin/bash
# round 3, session 19: bf16 -- C8 tiled-kernel variants (BK 64, write-after-barrier),
# C8 weight gradient with counted vmcnt waits + the workgroup-tiled C8 wgrad kernel,
# parallel BN-backward finalize.  Tests, A/B, family-2 retune, per-layer table.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_gpu_bf16.py -x -q > $O/s19_pytest_bf16.log 2>&1; echo pytest_bf16 rc=$?; tail -3 $O/s19_pytest_bf16.log
timeout 200 python -m pytest tests/test_gpu_layers.py -x -q -k "bn or norm or fused" > $O/s19_pytest_layers.log 2>&1; echo pytest_layers rc=$?; tail -2 $O/s19_pytest_layers.log
for k in wave tile; do
  LD_CONV_WGRAD_C8_KERNEL=$k timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline > $O/s19_step_bf16_wgrad_$k.log 2>&1
  echo "wgrad $k: $(grep 'ms/step' $O/s19_step_bf16_wgrad_$k.log)"
done
timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline > $O/s19_step_bf16_default.log 2>&1; echo "default: $(grep 'ms/step' $O/s19_step_bf16_default.log)"
# retune the C8 forward / dgrad family with the new candidates
timeout 420 python tools/tune_conv.py --fresh-family 2 --modes bf16 --out $O/s19_tune.txt > $O/s19_tune.log 2>&1; echo tune rc=$?; tail -1 $O/s19_tune.log
grep -c "sch1" $O/s19_tune.log; grep -c "bk64" $O/s19_tune.log; grep -c "ld_conv c8" $O/s19_tune.log
if [ -s $O/s19_tune.txt ]; then
  cp ld_amd/tune/gfx950.txt /tmp/gfx950_old.txt
  python - <<'PY'
import re
old=open('ld_amd/tune/gfx950.txt').read().split('\n')
hdr=[l for l in old if l.startswith('#')]
new=[l for l in open('gpurun_out/s19_tune.txt').read().split('\n') if l.strip() and not l.startswith('#')]
open('ld_amd/tune/gfx950.txt','w').write('\n'.join(hdr+new)+'\n')
print('table records', len(new))
PY
  timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline --layers $O/s19_layers_bf16.csv > $O/s19_step_bf16_retuned.log 2>&1; echo "retuned: $(grep 'ms/step' $O/s19_step_bf16_retuned.log)"; grep "conv total" $O/s19_step_bf16_retuned.log
fi
timeout 200 python tools/profile_step.py --mode fp32 --steps 15 --warmup 4 --pipeline > $O/s19_step_fp32.log 2>&1; echo "fp32: $(grep 'ms/step' $O/s19_step_fp32.log)"
