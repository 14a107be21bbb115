#!/bin/bash
# round 6, GPU session 6: which queue is the critical path of the bf16 / fp32 step now
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
for m in bf16 fp32; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06q_$m -o step -- python $R/tools/profile_step.py --mode $m --steps 8 --warmup 3 --pipeline > $R/$O/r06q_$m.log 2>&1)
t=$(find $O/r06q_$m -name '*kernel_trace.csv' | head -1)
python tools/queue_busy.py "$t" --steps 5 > $O/r06_queue_busy_$m.txt 2>&1; cat $O/r06_queue_busy_$m.txt | head -30
python tools/launches_per_step.py "$t" --steps 5 > $O/r06_launches_per_step_$m.txt 2>&1; head -3 $O/r06_launches_per_step_$m.txt
f=$(find $O/r06q_$m -name '*kernel_stats.csv' | head -1); cp "$f" $O/r06_rocprof_kernel_stats_$m.csv
rm -rf $O/r06q_$m
done
