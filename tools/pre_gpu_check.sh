#!/bin/bash
# Run before every gpurun call: the snapshot must carry a library that matches
# the Python bindings (a stale .so costs a whole GPU session).
set -e
cd "$(dirname "$0")/.."
python -m ld_amd.build > /dev/null
python - <<'PY'
import ast, glob, sys
from ld_amd import lib as L
L.get_lib()
for f in glob.glob('tools/*.py') + glob.glob('tests/*.py') + ['bench.py', '__graft_entry__.py']:
    ast.parse(open(f).read(), f)
print('pre_gpu_check: library loads, every binding resolves, scripts parse')
PY
