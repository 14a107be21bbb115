"""Where do the small device-to-device copies / ATen elementwise launches of a
train step come from?  One step under torch.profiler with Python stacks; prints
the call sites of aten::copy_, aten::add / add_, aten::fill_ / zero_, aten::cat.
    python tools/find_copies.py [fp32|bf16]     (through gpurun)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo, synthetic  # noqa: E402
from ld_amd.train import SGDTrainer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda:0')
Y.set_precision(mode)
det = model_zoo.build_seeded_ld_detector(50, 101, dev)
tr = SGDTrainer(det, lr=0.0025)
b = synthetic.synthetic_batch(2, (800, 1333), (800, 1344), 7, 1234)
d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
         gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
         gt_labels=[x.to(dev) for x in b['gt_labels']])
for _ in range(3):
    tr.step(d)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(d)
    torch.cuda.synchronize()
want = ('aten::copy_', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::cat',
        'aten::clone', 'aten::contiguous', 'aten::mul', 'aten::sum', 'aten::index_select')
sites = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        st = [s for s in (ev.stack or []) if '/ld_amd/' in s or 'bench.py' in s]
        where = ' <- '.join(s.split('/ld_amd/')[-1] for s in st[:3]) or '(autograd engine / no ld_amd frame)'
        sites[(ev.name, where)] += 1
for (name, where), n in sites.most_common(40):
    print(f'{n:4d}  {name:18s} {where}')
