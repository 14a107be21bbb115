#!/bin/bash
# round 3, GPU session 8: teacher prefetch across steps (now that the queue
# hits), PMC of a teacher 1x1 layer, conv HBM traffic of the whole step
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
for m in fp32 bf16; do
  timeout 200 python tools/profile_step.py --mode $m --steps 20 --warmup 5 2>/dev/null | grep img/s
  timeout 200 python tools/profile_step.py --mode $m --steps 20 --warmup 5 --pipeline 2>/dev/null | grep img/s
done | tee $O/s8_pipeline.txt
timeout 400 bash tools/pmc_run.sh l3c3_bn_res conv_stream -- python $R/tools/one_conv.py l3c3 fwd_bn_res > $O/s8_pmc_l3c3.log 2>&1; cat $O/pmc_l3c3_bn_res.txt
LD_ONE_CONV_REPS=50 timeout 100 python tools/one_conv.py l3c3 fwd_bn_res | tail -2
timeout 300 bash tools/pmc_traffic.sh l3c3_bn_res conv_stream -- python $R/tools/one_conv.py l3c3 fwd_bn_res > $O/s8_pmc_l3c3_traffic.log 2>&1; cat $O/pmc_traffic_l3c3_bn_res.txt
timeout 500 bash tools/pmc_traffic.sh convstep conv_ -- python $R/tools/profile_step.py --mode fp32 --steps 2 --warmup 1 > $O/s8_pmc_convstep.log 2>&1; cat $O/pmc_traffic_convstep.txt
