#!/bin/bash
# round 5, session 13: record the real backward's gradient-completion order (fixture
# of the 2-rank gloo arena test)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/record_backward_order.py gpurun_out/backward_order.json 2>&1 | tail -5
