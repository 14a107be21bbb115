#!/bin/bash
# round 6, GPU session 3: re-tune family 2 (C8 kernels) with the 8-wave LDS-DMA
# shapes among the candidates; bf16 test file; per-layer table and step time
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fused_block.py -q -m gpu -x > $O/s3_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s3_pytest.log
timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
timeout 1200 python tools/tune_conv.py --fresh-family 2 --modes bf16 --out $O/tune_r06.txt > $O/s3_tune.log 2>&1; echo tune rc=$?; tail -1 $O/s3_tune.log
grep -c "bk64 sch0" $O/s3_tune.log; grep "ld_conv c8" $O/s3_tune.log | awk '{print $NF, $0}' | sort -u -k2 | head -100 > $O/s3_tune_picks.txt; grep -E "256x192x8|256x256x8|128x256x8|256x128x8" $O/s3_tune.log | head -60
export LD_CONV_TUNE_FILE=$R/$O/tune_r06.txt
timeout 300 python tools/profile_step.py --mode bf16 --steps 6 --layers $O/layers_bf16_r06s3.csv > $O/s3_layers.log 2>&1; tail -2 $O/s3_layers.log
timeout 200 python tools/profile_step.py --mode bf16 --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
