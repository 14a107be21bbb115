"""Run one conv shape a few times in bf16 mode with a forced kernel shape (for
rocprofv3 --pmc runs):  python tools/one_conv_bf16.py <l2|head> <shape>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else 'l2'
if len(sys.argv) > 2:
    os.environ['LD_CONV_BF16_SHAPE'] = sys.argv[2]
from ld_amd import layers as Y  # noqa: E402

shapes = {
    'head': (2, 256, 256, 3, 1, 1, ((100, 168), (50, 84), (25, 42), (13, 21),
                                    (7, 11))),
    'l2': (2, 128, 128, 3, 1, 1, ((100, 168), )),
    'l3': (2, 256, 256, 3, 1, 1, ((50, 84), )),
}
N, cin, cout, k, s, p, levels = shapes[which]
dev = torch.device('cuda:0')
Y.set_precision('bf16')
P = sum(h * w for h, w in levels)
x = torch.randn(N, cin, P, device=dev)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
for _ in range(6):
    Y.conv_forward_raw(x, w, s, p, levels)
torch.cuda.synchronize()
print('done')
