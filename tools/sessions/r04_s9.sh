#!/bin/bash
# round 4, session 9: bucket all-reduce issued from the wgrad side stream -- A/B with forced collectives in a 1-rank RCCL group
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
export LD_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for rep in 1 2; do
for v in 0 1; do
  MASTER_PORT=$((29600 + rep * 10 + v)) LD_BUCKET_FROM_SIDE=$v timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/forced collectives, all-reduce from side stream=$v: /"
done
done
unset LD_FORCE_COLLECTIVES RANK WORLD_SIZE LOCAL_RANK
timeout 300 python tools/profile_step.py --mode fp32 --pipeline --steps 20 --warmup 5 2>&1 | grep "ms/step" | sed "s/^/no collectives: /"
timeout 600 python -m pytest tests/test_gpu_rccl.py -q -m gpu > $O/r04s9_pytest.log 2>&1; echo pytest rc=$?; tail -2 $O/r04s9_pytest.log
