#!/bin/bash
# round 6, GPU session 22: host profile of the eager bf16 step, backward included
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/host_profile.py bf16 20 > gpurun_out/s22_host_profile.txt 2>&1; head -75 gpurun_out/s22_host_profile.txt
