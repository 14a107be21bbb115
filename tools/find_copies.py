"""Where do the small copy / add launches of a train step come from?
torch.profiler with Python stacks, grouped by the innermost ld_amd frame."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import model_zoo  # noqa: E402
from ld_amd.train import SGDTrainer  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda:0')
det = model_zoo.build_seeded_ld_detector(50, 101, dev)
tr = SGDTrainer(det, lr=0.0025)
_, d = bench.make_batch(2, 7, 1234, dev)
for _ in range(3):
    tr.step(d)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(d)
torch.cuda.synchronize()
want = ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::add_',
        'aten::add', 'aten::to', 'aten::_to_copy', 'aten::zero_', 'aten::fill_',
        'aten::cat', 'aten::stack', 'aten::select_backward',
        'aten::slice_backward', 'aten::sum', 'aten::mul')
agg = collections.Counter()
for e in prof.events():
    if e.name not in want:
        continue
    frame = 'autograd/engine'
    for s in e.stack:
        if 'ld_amd' in s or 'bench.py' in s:
            frame = s.split('/')[-1][:70]
            break
    agg[(e.name, frame)] += 1
for (name, frame), n in agg.most_common(45):
    print(f'{n:5d}  {name:22s} {frame}')
