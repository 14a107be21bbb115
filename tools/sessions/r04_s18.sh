#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_rccl.py tests/test_gpu_e2e.py tests/test_gpu_bf16.py -q -m gpu -x > $O/r04s18_pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/r04s18_pytest.log | head -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04s18_bench.json 2> $O/r04s18_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04s18_bench.json").read().strip())
print("value",d["value"],d["ms_per_step"],"in-step",d["config"]["images_per_sec_teacher_in_step"])
print("bf16",d["bf16"]["value"],d["bf16"]["ms_per_step"])
print("graph",{m:(round(v.get("value",0),1),round(v.get("teacher_one_step_ahead",{}).get("value",0),1)) for m,v in d["hipgraph_step"].items()})
print("roofline",d["roofline"]["frac"], d["roofline"]["by_kind"]["conv_wgrad"]["tflops"])
PY
