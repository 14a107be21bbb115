#!/bin/bash
# rocprofv3 kernel statistics of the step (teacher one step ahead), fp32 + bf16
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
for m in fp32 bf16; do
  rm -rf /tmp/rp_$m
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/rp_$m -o st -- python $R/tools/profile_step.py --mode $m --steps 10 --warmup 3 --pipeline > $R/$O/s16_rocprof_$m.log 2>&1 )
  DB=$(find /tmp/rp_$m -name '*.db' | head -1)
  python tools/rocpd_stats.py $DB $O/rocprof_kernel_stats_$m.csv
  grep img/s $O/s16_rocprof_$m.log
done
