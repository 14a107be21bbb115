#!/bin/bash
# round 6: HBM-side traffic of the conv kernels of the serialised step, by kernel, on
# the final build (fp32 and bf16), and the counter attribution of the 8-wave LDS-DMA
# kernel on the head-tower shape
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
PMC_BY_KERNEL=1 timeout 900 tools/pmc_traffic.sh r06_conv_step_fp32_by_kernel "conv_" -- python $R/tools/profile_step.py --mode fp32 --serial --steps 4 --warmup 2 > $O/r6pmc_fp32.log 2>&1; head -8 $O/pmc_traffic_r06_conv_step_fp32_by_kernel.txt
PMC_BY_KERNEL=1 timeout 900 tools/pmc_traffic.sh r06_conv_step_bf16_by_kernel "conv_|bottleneck" -- python $R/tools/profile_step.py --mode bf16 --serial --steps 4 --warmup 2 > $O/r6pmc_bf16.log 2>&1; head -8 $O/pmc_traffic_r06_conv_step_bf16_by_kernel.txt
LD_CONV_C8_SHAPE=8x6x8x64 timeout 600 tools/pmc_run.sh r06_head_c8_t256 conv_t256 -- python $R/tools/one_conv_bf16.py head > $O/r6pmc_t256.log 2>&1; cat $O/pmc_r06_head_c8_t256.txt
grep -c fused_bottleneck $O/pmc_traffic_r06_conv_step_bf16_by_kernel.txt; grep "fused_bottleneck\|conv_stem" $O/pmc_traffic_r06_conv_step_*_by_kernel.txt | cut -c1-200
