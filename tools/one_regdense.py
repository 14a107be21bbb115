"""Run the train step's fused LD-KL + VLR-LD + Integral kernel
(loss_reg_dense_kernel via ld_loss_main_parts(LD_LOSS_PART_REG)) a few times at
the saturating size of bench.py's roofline_ldkl leg (2^24 anchor-side rows),
for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device('cuda:0')
launch, rows = bench._reg_dense_launcher(dev, [(2048, 2048)], [8], 1, 1.0)
for _ in range(6):
    launch()
torch.cuda.synchronize()
print('rows', rows, 'algorithmic bytes', rows * 207)
