"""Run one conv shape a few times (for rocprofv3 --pmc runs):
    python tools/one_conv.py <shape> [fwd|fwd_bn_res|dgrad|wgrad]
LD_CONV_STREAM / LD_CONV_WGRAD in the environment force a kernel shape."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import lib as L  # noqa: E402
from ld_amd import lossblock as LB  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'head'
kind = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
HEAD = ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))
shapes = {
    'head': (2, 256, 256, 3, 1, 1, HEAD),
    'l3': (2, 256, 256, 3, 1, 1, ((50, 84), )),
    'l3c3': (2, 256, 1024, 1, 1, 0, ((50, 84), )),
    'l3c1': (2, 1024, 256, 1, 1, 0, ((50, 84), )),
    'l2c3': (2, 128, 512, 1, 1, 0, ((100, 168), )),
    'cls': (2, 256, 80, 3, 1, 1, HEAD),
}
N, cin, cout, k, s, p, levels = shapes[which]
dev = torch.device('cuda:0')
P = sum(h * w for h, w in levels)
x = torch.randn(N, cin, P, device=dev)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
lib = L.get_lib()
d, out_levels = Y.conv_desc(N, cin, cout, k, k, s, p, levels)
go = torch.randn(N, cout, d.Pout, device=dev)
st = L.stream_ptr(dev)
if kind == 'fwd_bn_res':
    # the frozen teacher's bottleneck tail: conv -> folded BN -> + identity ->
    # ReLU in one launch (resnet._conv_bn / layers.conv_bn_act_infer)
    scale = torch.rand(cout, device=dev) + 0.5
    shift = torch.randn(cout, device=dev) * 0.1
    res = torch.randn(N, cout, d.Pout, device=dev)

    def run():
        Y.conv_forward_raw(x, w, s, p, levels, scale=scale, shift=shift,
                           residual=res, relu=True)
elif kind == 'fwd':
    def run():
        Y.conv_forward_raw(x, w, s, p, levels)
elif kind == 'dgrad':
    _, wt_bwd = Y.weight_images(w.requires_grad_(True), True)
    dx = torch.empty_like(x)

    def run():
        L.check(lib.ld_conv_dgrad(C.byref(d), L.ptr(go), L.ptr(wt_bwd),
                                  L.ptr(dx), st), 'dgrad')
else:
    dw = torch.empty_like(w)
    ws = LB.workspace(dev, lib.ld_conv_wgrad_workspace_bytes(C.byref(d)),
                      'wgrad')

    def run():
        L.check(lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(go), L.ptr(dw),
                                  0, L.ptr(ws), ws.numel(), st), 'wgrad')
for _ in range(6):
    run()
torch.cuda.synchronize()
reps = int(os.environ.get('LD_ONE_CONV_REPS', '0'))
if reps:  # event-timed average (the PMC runs leave this off)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flop = 2.0 * N * d.Pout * cin * cout * k * k
    print(f'{which} {kind} LD_CONV_WGRAD={os.environ.get("LD_CONV_WGRAD", "-")}: '
          f'{us:.1f} us, {flop / us / 1e6:.1f} TFLOP/s')
print('done')
