#!/bin/bash
# round 4, session 12: bench under torch.distributed.run (forced collectives, 1 rank) with the HW-queue default; stdout hygiene
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
LD_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s12_torchrun.out 2> $O/r04s12_torchrun.err; echo torchrun rc=$?
echo "stdout lines: $(wc -l < $O/r04s12_torchrun.out)"; head -c 300 $O/r04s12_torchrun.out; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s12_torchrun.out').read().strip())
print('forced collectives 1 rank:', d['value'], d['ms_per_step'], 'in-step', d['config']['images_per_sec_teacher_in_step'], 'rccl_ranks', d.get('rccl_ranks'))
PY
timeout 600 python bench.py --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s12_plain.out 2> $O/r04s12_plain.err; echo plain rc=$?; echo "stdout lines: $(wc -l < $O/r04s12_plain.out)"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s12_plain.out').read().strip())
print('plain:', d['value'], d['ms_per_step'])
PY
