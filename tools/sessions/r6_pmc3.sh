#!/bin/bash
# round 6: PMC traffic of the bf16 conv kernels on the build with the re-tuned C8 shape table
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
PMC_BY_KERNEL=1 timeout 900 tools/pmc_traffic.sh r06_conv_step_bf16_by_kernel "conv_|bottleneck" -- python $R/tools/profile_step.py --mode bf16 --serial --steps 4 --warmup 2 > $O/r6pmc_bf16.log 2>&1
python tools/pmc_conv_bytes.py $O/pmc_traffic_r06_conv_step_bf16_by_kernel.txt 7
