#!/bin/bash
# round 4, session 5: FETCH_SIZE calibration on 4 B / lane reads, conv traffic of the serial step, remaining GPU tests
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 bash tools/pmc_traffic.sh r04_calib_copy4B "copy_kernelILi1|copy_kernel<1" -- python /root/repo/tools/calib_fetch.py > $O/r04s5_c1.log 2>&1
timeout 300 bash tools/pmc_traffic.sh r04_calib_copy16B "copy_kernelILi4|copy_kernel<4" -- python /root/repo/tools/calib_fetch.py > $O/r04s5_c2.log 2>&1
cat $O/pmc_traffic_r04_calib_copy4B.txt $O/pmc_traffic_r04_calib_copy16B.txt
timeout 600 bash tools/pmc_traffic.sh r04_conv_step_fp32 "conv_" -- python /root/repo/tools/profile_step.py --mode fp32 --serial --steps 4 --warmup 2 > $O/r04s5_c3.log 2>&1
cat $O/pmc_traffic_r04_conv_step_fp32.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=4 > $O/r04s5_pytest.log 2>&1; echo pytest rc=$?; tail -15 $O/r04s5_pytest.log
