#!/bin/bash
# round 5, session 1 (diagnostics): L2 weight-stream probe, where the ATen / copy
# launches come from, host profile, hipGraph replay vs runtime queue knobs, the
# forced-collective 1-rank bench WITH the bf16 leg
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 200 python tools/probe/run_l2_weight_stream.py $O/r05_probe_l2_weight_stream.json > $O/r05s1_probe.log 2>&1; echo probe rc=$?
timeout 200 python tools/find_copies.py fp32 > $O/r05_find_copies_fp32.txt 2>&1; echo fc32 rc=$?
timeout 200 python tools/find_copies.py bf16 > $O/r05_find_copies_bf16.txt 2>&1; echo fcbf rc=$?
timeout 200 python tools/host_profile.py bf16 > $O/r05_host_profile_bf16.txt 2>&1; echo hp rc=$?
: > $O/r05_graph_queues.jsonl
run_gq() {  # hwq graphq wgrad_stream extra...
  ( export GPU_MAX_HW_QUEUES=$1; [ "$2" != - ] && export DEBUG_HIP_FORCE_GRAPH_QUEUES=$2; export LD_WGRAD_STREAM=$3
    timeout 240 python tools/graph_queues.py bf16 ${@:4} 2>>$O/r05s1_gq.err | tail -1 >> $O/r05_graph_queues.jsonl )
}
run_gq 4 - 1 --dot $O/r05_graph_bf16.dot
run_gq 8 - 1
run_gq 8 2 1
run_gq 8 1 1
run_gq 8 - 0
run_gq 4 2 1
run_gq 4 1 1
cat $O/r05_graph_queues.jsonl
LD_FORCE_COLLECTIVES=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $O/r05_bench_torchrun_1rank_forced_collectives_bf16.json 2> $O/r05s1_bench_fc.err; echo bench rc=$?
cat $O/r05_bench_torchrun_1rank_forced_collectives_bf16.json | head -c 1500
tail -5 $O/r05s1_probe.log; head -30 $O/r05_find_copies_bf16.txt
