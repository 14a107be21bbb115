#!/bin/bash
# round 6, GPU session 13: rocprofv3 kernel stats of the bf16 step after the lean
# norm backward changes (eager, pipeline), for the next choice of work
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06p_bf16 -o step -- python $R/tools/profile_step.py --mode bf16 --steps 8 --warmup 3 --pipeline > $R/$O/r06p.log 2>&1)
f=$(find $O/r06p_bf16 -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06p_bf16_kernel_stats.csv
t=$(find $O/r06p_bf16 -name '*kernel_trace.csv' | head -1)
python tools/queue_busy.py "$t" --steps 5 > $O/r06p_queue_busy_bf16.txt 2>&1
python tools/launches_per_step.py "$t" --steps 5 > $O/r06p_launches_bf16.txt 2>&1
rm -rf $O/r06p_bf16
tail -3 $O/r06p.log
