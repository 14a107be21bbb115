"""Timing attribution of the fp32 streaming forward kernel on the 50x84 layers of
the R101 teacher (LD_STREAM_DBG bit mask, results wrong by design): shape forced
to 1x1x1 d16 ks4 (what the shipped table picks there)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402

os.environ['LD_ALLOW_WRONG_RESULTS'] = '1'  # the variants timed here compute wrong results by design
dev = torch.device('cuda:0')
os.environ['LD_CONV_STREAM'] = '1x1x1x16x4'
NAMES = {0: 'full', 1: 'no ring refills (MFMA + epilogue)', 2: 'pixel loads read nothing',
         4: 'weight loads read nothing', 6: 'all loads read nothing', 8: 'no epilogue',
         9: 'MFMA only', 14: 'no traffic at all, ring issued'}
LAYERS = [('l3c3 256>1024 1x1 +bn+res+relu', 256, 1024, 1, 0, True),
          ('l3c1 1024>256 1x1 +bn+relu', 1024, 256, 1, 0, False),
          ('l3c2 256>256 3x3 +bn+relu', 256, 256, 3, 1, False)]
for name, cin, cout, k, pad, res in LAYERS:
    N, levels = 2, ((50, 84), )
    P = 50 * 84
    x = torch.randn(N, cin, P, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    scale = torch.rand(cout, device=dev) + 0.5
    shift = torch.randn(cout, device=dev) * 0.1
    r = torch.randn(N, cout, P, device=dev) if res else None
    flop = 2.0 * N * P * cin * cout * k * k

    def run():
        Y.conv_forward_raw(x, w, 1, pad, levels, scale=scale, shift=shift, residual=r,
                           relu=True)
    for dbg in (0, 1, 2, 4, 6, 8, 9, 14, 0):
        os.environ['LD_STREAM_DBG'] = str(dbg)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            best = us if best is None else min(best, us)
        print(f'{name:34s} dbg {dbg:2d} {NAMES[dbg]:36s} {best:7.1f} us {flop / best / 1e6:6.1f} TF',
              flush=True)
