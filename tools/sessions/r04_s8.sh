#!/bin/bash
# round 4, session 8: launcher refusal on a 1-GPU box, bench under torch.distributed.run with forced collectives, rccl + graph tests
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
python bench.py --gpus 2 --steps 2 > $O/r04s8_gpus2.out 2> $O/r04s8_gpus2.err; echo "bench --gpus 2 on a 1-GPU box: rc=$?"; tail -2 $O/r04s8_gpus2.err
LD_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline > $O/r04s8_bench_torchrun_1rank_forced.json 2> $O/r04s8_torchrun.err; echo torchrun rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s8_bench_torchrun_1rank_forced.json').read().strip().splitlines()[-1])
print('forced collectives 1 rank:', d['value'], d['ms_per_step'], 'rccl_ranks', d.get('rccl_ranks'), 'n_gpus', d['n_gpus'])
PY
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_graph.py -q -m gpu > $O/r04s8_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/r04s8_pytest.log
