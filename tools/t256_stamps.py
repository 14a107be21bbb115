"""Phase breakdown of the 8-wave LDS-DMA conv kernel from in-kernel clock stamps
(needs a library built with LD_BUILD_DEFS=-DLD_T256_STAMP):
    python tools/t256_stamps.py [head] [8x6x8x64]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import lib as L  # noqa: E402
from tools.bench_t256 import GEO  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'head'
    shape = sys.argv[2] if len(sys.argv) > 2 else '8x6x8x64'
    N, cin, cout, k, s, p, levels = GEO[which]
    dev = torch.device('cuda:0')
    Y.set_precision('bf16')
    Y.set_c8(True)
    P = sum(h * w for h, w in levels)
    x = torch.randn(N, cin, P, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k)**0.5
    x8 = Y.C8Act(Y.to_c8(x), x.shape)
    os.environ['LD_CONV_C8_SHAPE'] = shape
    lib = L.get_lib()
    nb = 4096
    buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        Y.conv_forward_raw(x8, w, s, p, levels, relu=True)
    torch.cuda.synchronize()
    assert lib.ld_debug_t256_stamps(C.c_void_p(buf.data_ptr())) == 0
    Y.conv_forward_raw(x8, w, s, p, levels, relu=True)
    torch.cuda.synchronize()
    assert lib.ld_debug_t256_stamps(C.c_void_p(0)) == 0
    full = buf.cpu().reshape(nb, 8, 16)
    live = full[:, 0, 0] != 0
    t = full[live][:, :, :5].double()
    wall = (full[live][:, :, 6] - full[live][:, :, 5]).double()  # 100 MHz ticks
    t0 = t[:, :, 0].min()
    print(f'{which} {shape}: {t.shape[0]} workgroups; clock ticks (s_memtime)')
    names = ['setup (entry -> first DMA issue)', 'first tile lands (DMA -> barrier)',
             'main loop', 'epilogue']
    for i, n in enumerate(names):
        d = t[:, :, i + 1] - t[:, :, i]
        print(f'  {n:36s} mean {d.mean():9.0f}  min {d.min():9.0f}  max {d.max():9.0f}')
    tot = t[:, :, 4] - t[:, :, 0]
    print(f'  {"wave lifetime":36s} mean {tot.mean():9.0f}  min {tot.min():9.0f}  max {tot.max():9.0f}')
    print(f'  wave lifetime on the 100 MHz wall clock: mean {wall.mean() / 100:.2f} us -> '
          f'{tot.mean() / (wall.mean() / 100) / 1e3:.3f} s_memtime ticks per ns')
    for i, n in ((8, 'in-loop wait: own fragment reads (lgkmcnt)'), (9, 'in-loop wait: own DMA (vmcnt)'),
                 (10, 'in-loop wait: barrier')):
        d = full[live][:, :, i].double()
        print(f'  {n:44s} mean {d.mean():9.0f}  min {d.min():9.0f}  max {d.max():9.0f}  (sum over the steps)')
    w5 = full[live][:, :, 5].double()
    w6 = full[live][:, :, 6].double()
    print(f'  wall clock: first wave start -> last wave end {(w6.max() - w5.min()) / 100:.2f} us; '
          f'workgroup starts spread over {(w5.amin(1).max() - w5.min()) / 100:.2f} us')


if __name__ == '__main__':
    main()
