#!/bin/bash
# round 6, GPU session 7: norm backward kernels (GroupNorm apply with a scalar
# statistics table, BN backward with every load up front): bit identity, layer
# tests, whole-step goldens, timing in the step
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_graph.py -q -m gpu -x -k "norm_backward or bn_act or gn_act or train_step or graphed_step_equals or fused_conv_bn" > $O/s7_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/s7_pytest.log
for m in fp32 bf16; do
for old in 1 0; do
echo "== $m LD_NN_OLD=$old"; LD_NN_OLD=$old timeout 200 python tools/profile_step.py --mode $m --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
done; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06n_fp32 -o step -- python $R/tools/profile_step.py --mode fp32 --steps 6 --warmup 3 --pipeline > $R/$O/r06n.log 2>&1)
f=$(find $O/r06n_fp32 -name '*kernel_stats.csv' | head -1); grep -E "gn_bwd|bn_act_bwd|gn_apply|gn_stats" "$f" | cut -c1-60,150-260 | head; rm -rf $O/r06n_fp32
