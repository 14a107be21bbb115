#!/bin/bash
# round 6, GPU session 2: the bf16 counters VERDICT r5 asked for (PMC of the shipped
# 4-wave tile kernel and of the 8-wave LDS-DMA kernel on the head-tower shape,
# by-kernel HBM traffic of the serialised bf16 step)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
LD_CONV_C8_SHAPE=4x4x4x32x1 timeout 600 tools/pmc_run.sh head_c8_tile128 conv_tile_c8 -- python $R/tools/one_conv_bf16.py head > $O/s2_pmc1.log 2>&1; cat $O/pmc_head_c8_tile128.txt
LD_CONV_C8_SHAPE=8x6x8x64 timeout 600 tools/pmc_run.sh head_c8_t256 conv_t256 -- python $R/tools/one_conv_bf16.py head > $O/s2_pmc2.log 2>&1; cat $O/pmc_head_c8_t256.txt
PMC_BY_KERNEL=1 timeout 900 tools/pmc_traffic.sh convstep_bf16 "conv_|bottleneck" -- python $R/tools/profile_step.py --mode bf16 --serial --steps 2 --warmup 1 > $O/s2_pmc3.log 2>&1; head -60 $O/pmc_traffic_convstep_bf16.txt
