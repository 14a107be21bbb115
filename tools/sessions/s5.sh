#!/bin/bash
# round 3, GPU session 5: graph path on real batches, config 4 at its shape,
# RCCL overlap trace, LD-KL PMC traffic, bench lines
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_config4.py tests/test_gpu_rccl.py tests/test_gpu_lossblock.py -q -m gpu -x --durations=8 > $O/s5_pytest.log 2>&1; echo pytest rc=$?; tail -4 $O/s5_pytest.log
# overlap trace: one rank, every collective through RCCL
( cd /tmp && RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 LD_FORCE_COLLECTIVES=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/ovl -o ovl -- python $GRAFT_REPO_ROOT/tools/profile_step.py --mode fp32 --steps 4 --warmup 3 > $GRAFT_REPO_ROOT/$O/s5_overlap_run.log 2>&1 )
DB=$(find /tmp/ovl -name '*.db' | head -1); echo db=$DB
[ -n "$DB" ] && python tools/overlap_trace.py $DB $O/overlap_fp32.txt > /dev/null 2>&1; tail -3 $O/overlap_fp32.txt
# LD-KL HBM traffic (separate PMC passes)
timeout 300 bash tools/pmc_traffic.sh regdense_r3 loss_reg_lean -- python tools/one_regdense.py > $O/s5_pmc.log 2>&1; tail -5 $O/pmc_traffic_regdense_r3.txt
# bench: default line, config 4, forced collectives
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_s5.json 2> $O/bench_s5.err; echo bench rc=$?
timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --no-bf16 --no-graph > $O/bench_s5_config4.json 2> $O/bench_s5_config4.err; echo bench4 rc=$?
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 LD_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-bf16 --no-graph --no-cpu-baseline --no-kernel-roofline > $O/bench_s5_forced.json 2> $O/bench_s5_forced.err; echo benchf rc=$?
python - <<'PY'
import json
for f in ('bench_s5','bench_s5_config4','bench_s5_forced'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().split('\n')[-1])
        print(f, round(d['value'],2), 'img/s', round(d['ms_per_step'],2),'ms', 'roof', round(d.get('roofline',{}).get('frac',0),3), 'ldkl', round(d.get('roofline_ldkl',{}).get('frac',0),3), round(d.get('roofline_ldkl',{}).get('frac_after_train_legs',0),3), 'bf16', round(d.get('bf16',{}).get('value',0),1), 'graph', {k:round(v.get('value',0),1) for k,v in d.get('hipgraph_step',{}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
