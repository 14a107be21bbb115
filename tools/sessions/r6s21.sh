#!/bin/bash
# round 6, GPU session 21: the shape-table key counts a C8 residual; re-tune of the C8
# kernel family on the lean / trunk-C8 data path; step-list timing of (a) old key,
# (b) new key + shipped table, (c) new key + re-tuned table
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1200 python tools/tune_conv.py --fresh-family 2 --modes bf16 --out $O/tune_r06b.txt > $O/s21_tune.log 2>&1; echo tune rc=$?; tail -1 $O/s21_tune.log
grep "ld_conv c8" $O/s21_tune.log | sort -u > $O/s21_tune_picks.txt; wc -l $O/s21_tune_picks.txt
for rep in 1 2; do
echo "== (a) old key"; LD_TUNE_KEY_RESC8=0 timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^pipelined_list" | cut -c1-120
echo "== (b) new key, shipped table"; timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^pipelined_list" | cut -c1-120
echo "== (c) new key, re-tuned table"; LD_CONV_TUNE_FILE=$R/$O/tune_r06b.txt timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^pipelined_list" | cut -c1-120
done
