"""Run the fused LD-KL + Integral kernel a few times at the saturating size
(for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import lossblock as LB  # noqa: E402

dev = torch.device('cuda:0')
rows = 1 << 22
s = torch.randn(68, rows, device=dev) * 3
t = torch.randn(68, rows, device=dev) * 3
w = torch.rand(rows, device=dev)
for _ in range(4):
    LB.kl_integral_dense(s, t, w, 10.0, 1.0, True)
torch.cuda.synchronize()
print('done')
