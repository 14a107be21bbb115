#!/bin/bash
# round 6, second half: HBM-side traffic of the bf16 conv kernels of the serialised
# step, by kernel, on the build with the lean norm paths (conv epilogues write the
# pre-affine result as a C8 image, the trainable trunk only C8 images)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
PMC_BY_KERNEL=1 timeout 900 tools/pmc_traffic.sh r06_conv_step_bf16_by_kernel "conv_|bottleneck" -- python $R/tools/profile_step.py --mode bf16 --serial --steps 4 --warmup 2 > $O/r6pmc_bf16.log 2>&1; head -8 $O/pmc_traffic_r06_conv_step_bf16_by_kernel.txt
python tools/pmc_conv_bytes.py $O/pmc_traffic_r06_conv_step_bf16_by_kernel.txt 7
# weight-gradient split heuristic under contention: the fixed-cost term (in 32-position
# steps) of ld_bf16_wgrad_c8_tile_splits; larger = fewer splits = less slab traffic
for fx in 6 12 24; do
echo "== LD_WGRAD_C8_FIXED=$fx"; LD_WGRAD_C8_FIXED=$fx timeout 400 python tools/bench_step_list.py bf16 40 2>&1 | grep -E "^pipelined_list" | cut -c1-160
done
