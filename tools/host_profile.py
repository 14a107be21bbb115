"""cProfile of the host side of the train step (find Python launch overhead)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import model_zoo  # noqa: E402
from ld_amd.train import SGDTrainer  # noqa: E402
import bench  # noqa: E402
from ld_amd import layers as Y  # noqa: E402

if len(sys.argv) > 1:
    Y.set_precision(sys.argv[1])  # fp32 | bf16

dev = torch.device('cuda:0')
det = model_zoo.build_seeded_ld_detector(50, 101, dev)
tr = SGDTrainer(det, lr=0.0025)
_, d = bench.make_batch(2, 7, 1234, dev)
for _ in range(3):
    tr.step(d)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(d)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(35)
