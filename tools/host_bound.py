"""Is the train step host-bound?  Same launch count at a tiny image size: the
step time there is (almost) pure host enqueue cost."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo, synthetic  # noqa: E402
from ld_amd.train import GraphedStep, SGDTrainer  # noqa: E402

if len(sys.argv) > 1:
    Y.set_precision(sys.argv[1])  # fp32 | bf16
dev = torch.device('cuda:0')
det = model_zoo.build_seeded_ld_detector(50, 101, dev)
tr = SGDTrainer(det, lr=0.0025)
for shape in ((256, 320), (800, 1344)):
    b = synthetic.synthetic_batch(2, shape, shape, [7, 7], 1234)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    for _ in range(4):
        tr.step(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(d)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(Y.get_precision(), shape,
          'enqueue %.1f ms/step, total %.1f ms/step' %
          ((t1 - t0) * 100, (t2 - t0) * 100), flush=True)
    g = GraphedStep(tr, d)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(Y.get_precision(), shape,
          'hipGraph replay: enqueue %.1f ms/step, total %.1f ms/step' %
          ((t1 - t0) * 100, (t2 - t0) * 100), flush=True)
    del g
