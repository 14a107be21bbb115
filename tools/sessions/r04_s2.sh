#!/bin/bash
# round 4, session 2: PMC of the new tiled wgrad vs the wave-private kernel vs the fwd kernel (head tower shape)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
LD_CONV_WGRAD_CFG=1,1,32,14,0 timeout 300 bash tools/pmc_run.sh r04s2_wgrad_tile_head conv_wgrad_tile_kernel -- python /root/repo/tools/one_conv.py head wgrad > $O/r04s2_last.log 2>&1; echo rc=$?
LD_CONV_WGRAD_CFG=1,2,32,14,0 timeout 300 bash tools/pmc_run.sh r04s2_wgrad_tile_head_kg2 conv_wgrad_tile_kernel -- python /root/repo/tools/one_conv.py head wgrad > $O/r04s2_last.log 2>&1; echo rc=$?
LD_CONV_WGRAD_CFG=0 timeout 300 bash tools/pmc_run.sh r04s2_wgrad_wave_head conv_wgrad_wave_kernel -- python /root/repo/tools/one_conv.py head wgrad > $O/r04s2_last.log 2>&1; echo rc=$?
timeout 300 bash tools/pmc_run.sh r04s2_fwd_head conv_stream_kernel -- python /root/repo/tools/one_conv.py head fwd > $O/r04s2_last.log 2>&1; echo rc=$?
LD_CONV_WGRAD_CFG=1,2,64,16,0 timeout 300 bash tools/pmc_run.sh r04s2_wgrad_tile_l3c1 conv_wgrad_tile_kernel -- python /root/repo/tools/one_conv.py l3c1 wgrad > $O/r04s2_last.log 2>&1; echo rc=$?
for f in $O/pmc_r04s2_*.txt; do echo "== $f"; cat $f; done
