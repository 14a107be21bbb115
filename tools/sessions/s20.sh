#!/bin/bash
# round 3, session 20: branch-free / grouped conv epilogues (fp32 stream, vec 1x1,
# bf16 kernels), then retune family 0 (fp32 stream) and 2 (bf16 C8) on the new code.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_layers.py tests/test_gpu_bf16.py -x -q > $O/s20_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s20_pytest.log
timeout 200 python tools/profile_step.py --mode fp32 --steps 15 --warmup 4 --pipeline > $O/s20_step_fp32_oldtable.log 2>&1; echo "fp32 old table: $(grep 'ms/step' $O/s20_step_fp32_oldtable.log)"
timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline > $O/s20_step_bf16_oldtable.log 2>&1; echo "bf16 old table: $(grep 'ms/step' $O/s20_step_bf16_oldtable.log)"
merge() {  # $1 = tuned table -> shipped table (keep the header)
  python - "$1" <<'PY'
import sys
old=open('ld_amd/tune/gfx950.txt').read().split('\n')
hdr=[l for l in old if l.startswith('#')]
new=[l for l in open(sys.argv[1]).read().split('\n') if l.strip() and not l.startswith('#')]
open('ld_amd/tune/gfx950.txt','w').write('\n'.join(hdr+new)+'\n')
print('table records', len(new))
PY
}
timeout 600 python tools/tune_conv.py --fresh-family 0 --modes fp32 --out $O/s20_tune_f0.txt > $O/s20_tune_f0.log 2>&1; echo tune0 rc=$?; tail -1 $O/s20_tune_f0.log
[ -s $O/s20_tune_f0.txt ] && merge $O/s20_tune_f0.txt
timeout 420 python tools/tune_conv.py --fresh-family 2 --modes bf16 --out $O/s20_tune_f2.txt > $O/s20_tune_f2.log 2>&1; echo tune2 rc=$?; tail -1 $O/s20_tune_f2.log
[ -s $O/s20_tune_f2.txt ] && merge $O/s20_tune_f2.txt
cp ld_amd/tune/gfx950.txt $O/s20_gfx950.txt
timeout 300 python tools/profile_step.py --mode fp32 --steps 15 --warmup 4 --pipeline --layers $O/s20_layers_fp32.csv > $O/s20_step_fp32.log 2>&1; echo "fp32 retuned: $(grep 'ms/step' $O/s20_step_fp32.log)"; grep "conv total" $O/s20_step_fp32.log | tail -1
timeout 300 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline --layers $O/s20_layers_bf16.csv > $O/s20_step_bf16.log 2>&1; echo "bf16 retuned: $(grep 'ms/step' $O/s20_step_bf16.log)"; grep "conv total" $O/s20_step_bf16.log | tail -1
