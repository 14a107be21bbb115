#!/bin/bash
# round 6, GPU session 1: first run of the 8-wave LDS-DMA C8 kernel (bit identity,
# timing on the head tower) + the bf16 counters VERDICT r5 asked for (PMC of the
# shipped 4-wave tile kernel on the head-tower shape, by-kernel HBM traffic of the
# serialised bf16 step)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x -k "c8_kernel_bit_identical or c8_only_output" > $O/s1_pytest.log 2>&1; echo pytest rc=$?; tail -5 $O/s1_pytest.log
for g in head fpn l3; do timeout 300 python tools/bench_t256.py $g $O/t256_$g.json 2>&1 | tail -8; done
LD_CONV_C8_SHAPE=4x4x4x32x1 timeout 600 tools/pmc_run.sh head_c8_tile128 conv_tile_c8 -- python tools/one_conv_bf16.py head > /dev/null 2>&1; cat $O/pmc_head_c8_tile128.txt
LD_CONV_C8_SHAPE=8x6x8x64 timeout 600 tools/pmc_run.sh head_c8_t256 conv_t256 -- python tools/one_conv_bf16.py head > /dev/null 2>&1; cat $O/pmc_head_c8_t256.txt
PMC_BY_KERNEL=1 timeout 900 tools/pmc_traffic.sh convstep_bf16 "conv_|bottleneck" -- python tools/profile_step.py --mode bf16 --serial --steps 2 --warmup 1 > /dev/null 2>&1; head -50 $O/pmc_traffic_convstep_bf16.txt
