"""Timing attribution of the tiled fp32 wgrad main loop (head-tower shape): the
kernel with parts removed (LD_WGRAD_DBG bit mask, results wrong by design)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import lib as L  # noqa: E402

HEAD = ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))
os.environ['LD_ALLOW_WRONG_RESULTS'] = '1'  # the variants timed here compute wrong results by design
dev = torch.device('cuda:0')
lib = L.get_lib()
d, _ = Y.conv_desc(2, 256, 256, 3, 3, 1, 1, HEAD)
x = torch.randn(2, 256, d.Pin, device=dev)
dy = torch.randn(2, 256, d.Pout, device=dev)
dw = torch.empty(256, 256, 3, 3, device=dev)
ws = torch.zeros(lib.ld_conv_tune_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
st = L.stream_ptr(dev)
flop = 2.0 * 2 * d.Pout * 256 * 256 * 9
NAMES = {0: 'full', 1: '-barrier', 2: '-gloads', 4: '-ldswrites', 6: '-gloads-ldswrites',
         8: '-decode', 14: '-gloads-ldswrites-decode', 15: '-all but frag reads',
         16: '-fragreads', 30: 'mfma+barrier only', 31: 'mfma only',
         32: 'x4 loads (timing only)', 33: 'x4 loads -barrier',
         64: 'loads two slices ahead', 128: 'X loads read nothing'}
if os.environ.get('WGRAD_DBG_TAP3'):
    os.environ['LD_CONV_WGRAD_CFG'] = '2,0,0,21,0'
    for dbg in (0, 1, 0, 1):
        os.environ['LD_WGRAD_DBG'] = str(dbg)

        def run():
            L.check(lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), 0,
                                      L.ptr(ws), ws.numel(), st), 'wgrad')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            best = us if best is None else min(best, us)
        print(f'tap3 dbg {dbg} ({"only group 0 reads X" if dbg else "full"}) {best:8.1f} us '
              f'{flop / best / 1e6:6.1f} TF', flush=True)
    sys.exit(0)
for splits in (14, ):
    os.environ['LD_CONV_WGRAD_CFG'] = f'1,1,32,{splits},0'
    for dbg in (0, 64, 128, 0, 64, 128, 2):
        os.environ['LD_WGRAD_DBG'] = str(dbg)

        def run():
            L.check(lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), 0,
                                      L.ptr(ws), ws.numel(), st), 'wgrad')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            best = us if best is None else min(best, us)
        print(f'splits {splits} dbg {dbg:2d} {NAMES[dbg]:28s} {best:8.1f} us '
              f'{flop / best / 1e6:6.1f} TF', flush=True)
