#!/bin/bash
# round 6, GPU session 5: bench.py under torch.distributed.run with one rank and the
# collectives forced through RCCL (the multi-process code path on a one-GPU box)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
LD_FORCE_COLLECTIVES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_s5_forced.json 2> $O/bench_s5_forced.err; echo rc=$?; tail -3 $O/bench_s5_forced.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s5_forced.json').read().strip().split('\n')[-1])
c=d['config']
print('forced-collective 1-rank: value', round(d['value'],2), 'bracket', round(d['images_per_sec_k_step_bracket'],2), 'rccl_ranks', d['rccl_ranks'], 'pin', c['rank_cpu_placement'], 'exposed', c['exposed_allreduce_ms_per_step'], 'per-rank', c['ms_per_step_per_rank_k_step_bracket'])
print('bf16', round(d['bf16']['value'],2), round(d['bf16']['images_per_sec_k_step_bracket'],2))
PY
