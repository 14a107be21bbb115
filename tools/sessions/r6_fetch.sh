#!/bin/bash
# round 6: read amplification of the teacher's 1 x 1 convs: FETCH_SIZE and
# time per launch of one layer under different streaming tile shapes
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=$GRAFT_REPO_ROOT
for layer in l3c1 l3c3; do
for shape in "" 1x1x1x16x4 2x1x1x16x4 1x2x1x16x4 1x1x1x16x1 2x2x1x8x1 2x2x2x8x1 1x1x4x16x1 2x1x2x16x1; do
  if [ -n "$shape" ]; then export LD_CONV_STREAM=$shape; else unset LD_CONV_STREAM; fi
  t=$(LD_ONE_CONV_REPS=50 python $R/tools/one_conv.py $layer fwd_bn_res 2>/dev/null | grep TFLOP)
  tools/pmc_traffic.sh f_${layer}_$shape conv_stream -- python $R/tools/one_conv.py $layer fwd_bn_res > /dev/null 2>&1
  f=$(grep FETCH_SIZE $O/pmc_traffic_f_${layer}_$shape.txt | head -1 | awk '{print $2}')
  w=$(grep WRITE_SIZE $O/pmc_traffic_f_${layer}_$shape.txt | head -1 | awk '{print $2}')
  echo "$layer shape=[$shape] $t  FETCH_KiB=$f (x2 = bytes/1024)  WRITE_KiB=$w"
  rm -f $O/pmc_traffic_f_${layer}_$shape.txt
done; done
