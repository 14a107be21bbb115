"""Which host code issues the ~100 `__amd_rocclr_copyBuffer` / fill launches of a
train step?  One step under torch.profiler (CPU + device activities, Python
stacks), exported as a chrome trace; every runtime memcpy / memset call is
attributed to the innermost enclosing ld_amd / torch frame by time containment.
    python tools/find_memcpy.py [fp32|bf16]     (through gpurun)"""
import collections
import json
import os
import sys
import tempfile

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo  # noqa: E402
from ld_amd.train import SGDTrainer  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda:0')
Y.set_precision(mode)
det = model_zoo.build_seeded_ld_detector(50, 101, dev)
tr = SGDTrainer(det, lr=0.0025)
_, d0 = bench.make_batch(2, 7, 1234, dev)
_, d1 = bench.make_batch(2, 7, 5678, dev)
for i in range(4):
    tr.step(d0 if i % 2 else d1, next_data=d1 if i % 2 else d0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA],
             with_stack=True) as prof:
    tr.step(d0, next_data=d1)
    torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), 'trace.json')
prof.export_chrome_trace(path)
ev = json.load(open(path))['traceEvents']
py = [e for e in ev if e.get('cat') == 'python_function' and 'dur' in e]
ops = [e for e in ev if e.get('cat') == 'cpu_op' and 'dur' in e]
rt = [e for e in ev if e.get('cat') in ('cuda_runtime', 'cuda_driver')
      and any(k in e.get('name', '') for k in ('Memcpy', 'Memset', 'memcpy', 'memset'))]
kern = collections.Counter(e['name'][:60] for e in ev
                           if e.get('cat') in ('kernel', 'gpu_memcpy', 'gpu_memset'))
print('device-side events of the step (top 12 by count):')
for n, c in kern.most_common(12):
    print(f'  {c:5d}  {n}')
print(f'runtime memcpy/memset calls in the step: {len(rt)}')
sites = collections.Counter()
for r in rt:
    t, tid = r['ts'], r.get('tid')
    inner_op = None
    for o in ops:
        if o.get('tid') == tid and o['ts'] <= t <= o['ts'] + o['dur']:
            if inner_op is None or o['dur'] < inner_op['dur']:
                inner_op = o
    frames = [p for p in py if p.get('tid') == tid and p['ts'] <= t <= p['ts'] + p['dur']]
    frames.sort(key=lambda p: p['dur'])
    mine = [p['name'] for p in frames if 'ld_amd' in p['name'] or 'bench.py' in p['name']][:2]
    sites[(r['name'], inner_op['name'] if inner_op else '-',
           ' <- '.join(m.split('ld_amd/')[-1] for m in mine) or '(no ld_amd frame)')] += 1
for (name, op, where), n in sites.most_common(40):
    print(f'{n:4d}  {name:22s} {op:28s} {where}')
