"""Kernel micro-benchmarks on the MI355X (run through gpurun).  Writes
gpurun_out/kernels_<tag>.json.

  * the north-star fused LD-KL + Integral kernel at a saturating size
    (2^22 anchors = 2^24 anchor-side rows) for vector widths 1/2/4, fwd-only
    and fused fwd+grad, vs the HBM roofline;
  * the fp32 MFMA implicit-GEMM conv (fwd / dgrad / wgrad) on the layer shapes
    of the C2 config, next to MIOpen (torch F.conv2d) on the same shapes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from ld_amd import layers as Y  # noqa: E402
from ld_amd import lossblock as LB  # noqa: E402


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True),
            torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3  # median seconds


def bench_kl(out):
    dev = torch.device('cuda:0')
    res = []
    for rows in (1 << 22, 44800):  # saturating, and the C2 batch size
        s = torch.randn(68, rows, device=dev) * 3
        t = torch.randn(68, rows, device=dev) * 3
        w = torch.rand(rows, device=dev)
        for vec in (1, 2, 4):
            for nt in (0, 1):
                os.environ['LD_KL_VEC'] = str(vec)
                os.environ['LD_KL_NT'] = str(nt)
                for grad in (False, True):
                    dt = timeit(lambda: LB.kl_integral_dense(s, t, w, 10.0,
                                                             1.0, grad))
                    b = rows * 4 * (148 + (68 if grad else 0))
                    res.append(dict(rows_side=rows * 4, vec=vec, nt=nt,
                                    grad=grad, us=round(dt * 1e6, 1),
                                    GBps=round(b / dt / 1e9),
                                    frac_of_8TBps=round(b / dt / 8e12, 3)))
                    print(res[-1], flush=True)
    os.environ.pop('LD_KL_VEC', None)
    os.environ.pop('LD_KL_NT', None)
    out['kl_integral_dense'] = res


CONV_SHAPES = [
    # name, N, Cin, Cout, k, stride, pad, levels   (C2: 2 x 800 x 1344)
    ('r50.l1.conv2 3x3 64', 2, 64, 64, 3, 1, 1, ((200, 336), )),
    ('r50.l1.conv3 1x1 64-256', 2, 64, 256, 1, 1, 0, ((200, 336), )),
    ('r50.l2.conv2 3x3 128', 2, 128, 128, 3, 1, 1, ((100, 168), )),
    ('r50.l2.conv3 1x1 128-512', 2, 128, 512, 1, 1, 0, ((100, 168), )),
    ('r50.l3.conv2 3x3 256', 2, 256, 256, 3, 1, 1, ((50, 84), )),
    ('r50.l3.conv1 1x1 1024-256', 2, 1024, 256, 1, 1, 0, ((50, 84), )),
    ('r50.l4.conv2 3x3 512', 2, 512, 512, 3, 1, 1, ((25, 42), )),
    ('r50.l3.0.conv2 3x3 s2 256', 2, 256, 256, 3, 2, 1, ((100, 168), )),
    ('fpn.out 3x3 256 P3', 2, 256, 256, 3, 1, 1, ((100, 168), )),
    ('head tower 3x3 256 all levels', 2, 256, 256, 3, 1, 1,
     ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))),
    ('head gfl_cls 3x3 256-80', 2, 256, 80, 3, 1, 1,
     ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))),
    ('r101.l3.conv3 1x1 256-1024', 2, 256, 1024, 1, 1, 0, ((50, 84), )),
    ('r50.l4.conv1 1x1 2048-512', 2, 2048, 512, 1, 1, 0, ((25, 42), )),
    ('r50.l4.conv3 1x1 512-2048', 2, 512, 2048, 1, 1, 0, ((25, 42), )),
    ('r50.l2.conv1 1x1 512-128', 2, 512, 128, 1, 1, 0, ((100, 168), )),
    ('r50.l1.conv1 1x1 256-64', 2, 256, 64, 1, 1, 0, ((200, 336), )),
]

TILES = ['128x128x16', '128x64x16', '64x128x16', '64x64x32', '64x64x16',
         '128x128x32', '128x64x32', '64x128x32']


def bench_conv_tiles(out):
    """Forward implicit-GEMM under every tile shape (LD_CONV_TILE)."""
    dev = torch.device('cuda:0')
    res = []
    for name, N, cin, cout, k, stride, pad, levels in CONV_SHAPES:
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        y, _ = Y.conv_forward_raw(x, w, stride, pad, levels)
        flops = 2.0 * N * y.shape[2] * cout * cin * k * k
        r = dict(name=name, J=N * y.shape[2], cout=cout, K=cin * k * k)
        os.environ.pop('LD_CONV_TILE', None)
        t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels))
        r['auto'] = round(flops / t / 1e12, 1)
        for tile in ('64x64x32', '32x64x32', '32x64x16'):
            os.environ['LD_CONV_TILE'] = tile
            os.environ['LD_CONV_KG'] = '1'
            t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels))
            r[tile] = round(flops / t / 1e12, 1)
        os.environ.pop('LD_CONV_KG', None)
        os.environ.pop('LD_CONV_TILE', None)
        for tile in TILES if os.environ.get('LD_SWEEP_TILES') else []:
            if cout <= 64 and tile.startswith('128'):
                continue
            os.environ['LD_CONV_TILE'] = tile
            t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels))
            r[tile] = round(flops / t / 1e12, 1)
        os.environ.pop('LD_CONV_TILE', None)
        res.append(r)
        print(r, flush=True)
    out['conv_tiles'] = res



STREAM_CFGS = ['2x2x2x8x1', '2x2x1x8x1', '2x1x2x8x1', '1x2x2x8x1', '1x2x1x8x1',
               '1x1x2x8x1', '1x1x1x8x1', '1x1x4x8x1', '3x1x1x8x1', '3x2x1x8x1',
               '1x1x1x8x4', '1x2x1x8x4', '2x1x1x8x4', '2x2x1x8x4',
               # 16-deep ring
               '1x1x1x16x1', '1x1x2x16x1', '1x1x4x16x1', '2x1x2x16x1',
               '1x2x2x16x1', '2x1x1x16x1', '1x1x1x16x4', '2x1x1x16x4',
               '1x2x1x16x4']


def bench_stream_sweep(out):
    """Forward conv under every streaming-kernel shape (LD_CONV_STREAM) next to
    the LDS kernel (LD_CONV_STREAM=0)."""
    dev = torch.device('cuda:0')
    res = []
    for name, N, cin, cout, k, stride, pad, levels in CONV_SHAPES:
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        y, _ = Y.conv_forward_raw(x, w, stride, pad, levels)
        flops = 2.0 * N * y.shape[2] * cout * cin * k * k
        r = dict(name=name, J=N * y.shape[2], cout=cout, K=cin * k * k)
        os.environ['LD_CONV_STREAM'] = '0'
        t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels), 2, 5)
        r['lds'] = round(flops / t / 1e12, 1)
        for cfg in STREAM_CFGS:
            tm, tn, wvm = (int(v) for v in cfg.split('x')[:3])
            if wvm * tm * 32 >= 2 * cout and wvm > 1 and not cfg.endswith('x4'):
                continue  # more than half the workgroup's rows would be padding
            if int(cfg.split('x')[3]) == 16 and wvm and cin % 32:
                continue
            os.environ['LD_CONV_STREAM'] = cfg
            t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels), 2, 5)
            r[cfg] = round(flops / t / 1e12, 1)
        os.environ.pop('LD_CONV_STREAM', None)
        t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels), 2, 5)
        r['auto'] = round(flops / t / 1e12, 1)
        res.append(r)
        best = max((v, k_) for k_, v in r.items()
                   if isinstance(v, float) and k_ not in ('auto', ))
        print(name, '| best', best, '|',
              ' '.join(f'{k_}={v}' for k_, v in r.items()
                       if isinstance(v, float)), flush=True)
    out['conv_stream'] = res


def bench_conv(out, with_miopen=True):
    dev = torch.device('cuda:0')
    res = []
    for name, N, cin, cout, k, stride, pad, levels in CONV_SHAPES:
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        y, ol = Y.conv2d(xg, wg, None, stride, pad, levels)
        Pout = y.shape[2]
        go = torch.randn_like(y)
        flops = 2.0 * N * Pout * cout * cin * k * k
        r = dict(name=name, gflop=flops / 1e9)

        t_f = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels))
        import ctypes as C
        from ld_amd import lib as L
        lib = L.get_lib()
        d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
        _, wt_bwd = Y.weight_images(wg, True)
        dx = torch.empty_like(x)
        st = L.stream_ptr(dev)
        t_d = timeit(lambda: lib.ld_conv_dgrad(C.byref(d), L.ptr(go),
                                               L.ptr(wt_bwd), L.ptr(dx), st))
        dw = torch.empty_like(w)
        need = lib.ld_conv_wgrad_workspace_bytes(C.byref(d))
        ws = LB.workspace(dev, need, 'wgrad')
        t_w = timeit(lambda: lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(go),
                                               L.ptr(dw), 0, L.ptr(ws),
                                               ws.numel(), st))
        r.update(hip_fwd_ms=t_f * 1e3, hip_fwd_tflops=flops / t_f / 1e12,
                 hip_dgrad_ms=t_d * 1e3, hip_dgrad_tflops=flops / t_d / 1e12,
                 hip_wgrad_ms=t_w * 1e3, hip_wgrad_tflops=flops / t_w / 1e12)
        if with_miopen and len(levels) == 1:
            h, wd = levels[0]
            x4 = x.reshape(N, cin, h, wd).clone().requires_grad_(True)
            w4 = w.clone().requires_grad_(True)
            torch.backends.cudnn.benchmark = True
            t_mf = timeit(lambda: F.conv2d(x4.detach(), w4.detach(), None,
                                           stride, pad))
            y4 = F.conv2d(x4, w4, None, stride, pad)
            go4 = torch.randn_like(y4)

            def bwd():
                torch.autograd.grad(y4, (x4, w4), go4, retain_graph=True)

            t_mb = timeit(bwd)
            r.update(miopen_fwd_ms=t_mf * 1e3,
                     miopen_fwd_tflops=flops / t_mf / 1e12,
                     miopen_bwd_ms=t_mb * 1e3,
                     miopen_bwd_tflops=2 * flops / t_mb / 1e12)
        res.append(r)
        print({k: (round(v, 3) if isinstance(v, float) else v)
               for k, v in r.items()}, flush=True)
    out['conv'] = res


BF16_CFGS = ['2x2x2x2x1', '2x2x1x2x1', '2x1x2x4x1', '1x2x2x2x1', '1x1x2x4x1',
             '1x1x1x4x1', '1x1x4x4x1', '2x1x4x4x1', '1x1x1x4x4', '2x1x1x4x4',
             '1x2x1x2x4', '2x2x1x2x4', '4x4x0x32x2', '4x4x0x32x4',
             '2x4x0x32x4', '4x2x0x32x4', '4x2x0x32x2']


def bench_bf16(out):
    """bf16-MFMA conv (conv_bf16.hip) per C2 layer shape: forward under every
    streaming shape (LD_CONV_BF16_SHAPE), then forward / dgrad / wgrad with the
    library's own pick, next to the fp32 kernels on the same tensors."""
    import ctypes as C
    from ld_amd import lib as L
    lib = L.get_lib()
    dev = torch.device('cuda:0')
    res = []
    for name, N, cin, cout, k, stride, pad, levels in CONV_SHAPES:
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, device=dev)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).requires_grad_(True)
        d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
        flops = 2.0 * N * d.Pout * cout * cin * k * k
        r = dict(name=name, J=N * d.Pout, cout=cout, K=cin * k * k,
                 gflop=flops / 1e9)
        go = torch.randn(N, cout, d.Pout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        st = L.stream_ptr(dev)
        ws = LB.workspace(dev, lib.ld_conv_wgrad_workspace_bytes(C.byref(d)),
                          'wgrad')
        for mode in ('fp32', 'bf16'):
            Y.set_precision(mode)
            bf = mode == 'bf16'
            if bf and (cin % 16 or cout % 16):
                continue
            t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels), 2, 7)
            r[mode + '_fwd'] = round(flops / t / 1e12, 1)
            _, wt_bwd = Y.weight_images(w, True, bf16=bf, need_fwd=False)
            dg = lib.ld_conv_bf16_dgrad if bf else lib.ld_conv_dgrad
            t = timeit(lambda: dg(C.byref(d), L.ptr(go), L.ptr(wt_bwd), L.ptr(dx),
                                  st), 2, 7)
            r[mode + '_dgrad'] = round(flops / t / 1e12, 1)
            wg = lib.ld_conv_bf16_wgrad if bf else lib.ld_conv_wgrad
            t = timeit(lambda: wg(C.byref(d), L.ptr(x), L.ptr(go), L.ptr(dw), 0,
                                  L.ptr(ws), ws.numel(), st), 2, 7)
            r[mode + '_wgrad'] = round(flops / t / 1e12, 1)
        Y.set_precision('bf16')
        if cin % 16 == 0:
            steps = cin // 16
            for cfg in BF16_CFGS:
                dd = int(cfg.split('x')[3])
                if cfg.split('x')[2] == '0':
                    if cin % 32:
                        continue
                elif steps % dd:
                    continue
                os.environ['LD_CONV_BF16_SHAPE'] = cfg
                t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels),
                           2, 5)
                r[cfg] = round(flops / t / 1e12, 1)
            os.environ.pop('LD_CONV_BF16_SHAPE', None)
        Y.set_precision('fp32')
        res.append(r)
        print(' '.join(f'{k_}={v}' for k_, v in r.items()), flush=True)
    out['conv_bf16'] = res


def bench_c8(out):
    """bf16 forward / dgrad per C2 layer shape: fp32-input kernels (shape table
    / model pick) vs the C8-input tiled kernel under every tile shape, with and
    without the conversion launch."""
    from ld_amd import lib as L
    import ctypes as C
    dev = torch.device('cuda:0')
    lib = L.get_lib()
    Y.set_precision('bf16')
    Y._C8_ALL = True
    shapes = ['4x4x2', '4x4x4', '2x4x4', '4x2x4', '2x2x4', '2x4x2', '4x2x2',
              '2x2x2', '4x8x2', '2x8x2']
    res = []
    for name, N, cin, cout, k, stride, pad, levels in CONV_SHAPES:
        if cin % 32 or cout % 32:
            continue
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
        flops = 2.0 * N * d.Pout * cout * cin * k * k
        r = dict(name=name)
        Y.set_c8(False)
        t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels), 2, 7)
        r['fp32in'] = round(flops / t / 1e12, 1)
        Y.set_c8(True)
        xc = Y.to_c8(x)
        t = timeit(lambda: lib.ld_conv_to_c8(L.ptr(x), N, cin, P, L.ptr(xc),
                                             L.stream_ptr(dev)), 2, 7)
        r['to_c8_us'] = round(t * 1e6, 1)
        r['to_c8_GBps'] = round(x.numel() * 6 / t / 1e9)
        best = 0
        for sh in shapes:
            os.environ['LD_CONV_C8_SHAPE'] = sh
            t = timeit(lambda: Y.conv_forward_raw(x, w, stride, pad, levels),
                       2, 7)  # cached image: the GEMM alone
            r[sh] = round(flops / t / 1e12, 1)
            best = max(best, r[sh])
        os.environ.pop('LD_CONV_C8_SHAPE', None)
        r['best_c8'] = best
        tb = flops / (best * 1e12)
        r['best_c8_incl_convert'] = round(
            flops / (tb + r['to_c8_us'] * 1e-6) / 1e12, 1)
        res.append(r)
        print(r, flush=True)
    Y.set_precision('fp32')
    out['conv_c8'] = res


def bench_wgrad(out):
    """Weight gradient per C2 layer shape under each kernel (LD_CONV_WGRAD:
    32 / 16 positions per step of the wave-private kernel); the
    slab reduce pass is included in every number."""
    import ctypes as C
    from ld_amd import lib as L
    dev = torch.device('cuda:0')
    lib = L.get_lib()
    res = []
    for name, N, cin, cout, k, stride, pad, levels in CONV_SHAPES + [
            ('head gfl_reg 3x3 256-68', 2, 256, 68, 3, 1, 1,
             ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))),
            ('r50.l2.0.conv1 1x1 256-128', 2, 256, 128, 1, 1, 0, ((200, 336), )),
            ('r50.l3.0.ds 1x1 s2 512-1024', 2, 512, 1024, 1, 2, 0, ((100, 168), ))]:
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, device=dev)
        d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
        go = torch.randn(N, cout, d.Pout, device=dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        flops = 2.0 * N * d.Pout * cout * cin * k * k
        st = L.stream_ptr(dev)
        r = dict(name=name)
        for mode in ('32', '16'):
            os.environ['LD_CONV_WGRAD'] = mode
            need = lib.ld_conv_wgrad_workspace_bytes(C.byref(d))
            ws = LB.workspace(dev, need, 'wgrad')
            t = timeit(lambda: lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(go),
                                                 L.ptr(dw), 0, L.ptr(ws),
                                                 ws.numel(), st), 2, 7)
            r['wgrad' + mode] = round(flops / t / 1e12, 1)
        os.environ.pop('LD_CONV_WGRAD', None)
        res.append(r)
        print(r, flush=True)
    out['conv_wgrad'] = res


def bench_pipeline(out):
    """Device input pipeline at the C2 batch geometry (2 COCO-sized images ->
    2 x 3 x 800 x 1088 fp32): whole call (pinned staging + H2D + kernel) and
    the kernel alone on device-resident raw images, against its HBM bytes
    (3 B read per source pixel touched + 12 B written per padded pixel)."""
    import ctypes as C
    from ld_amd import lib as L
    from ld_amd.pipeline import DevicePipeline
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, (480, 640, 3)).astype(np.uint8),
            rs.randint(0, 256, (427, 640, 3)).astype(np.uint8)]
    pipe = DevicePipeline(device='cuda:0')
    np.random.seed(0)
    plans = pipe.plan([im.shape[:2] for im in imgs])
    t_call = timeit(lambda: pipe(imgs, plans=plans))
    res = pipe(imgs, plans=plans)
    N, _, Hp, Wp = res['img'].shape
    lib = L.get_lib()
    raw = [torch.from_numpy(im).cuda() for im in imgs]
    desc = (L.ImageT * N)()
    for i, (p, im) in enumerate(zip(plans, imgs)):
        desc[i].data = raw[i].data_ptr()
        desc[i].src_h, desc[i].src_w = im.shape[:2]
        desc[i].new_h, desc[i].new_w = p['img_shape'][:2]
        desc[i].flip = int(p['flip'])
    dd = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).cuda()
    o = torch.empty_like(res['img'])
    mean = (C.c_float * 3)(*pipe.mean.tolist())
    sinv = (C.c_float * 3)(*pipe.std_inv.tolist())
    st = torch.cuda.current_stream().cuda_stream

    def k():
        L.check(lib.ld_preprocess_batch(dd.data_ptr(), N, Hp, Wp, mean, sinv, 1,
                                        o.data_ptr(), st), 'pre')
    t_k = timeit(k, iters=50)
    nbytes = sum(im.size for im in imgs) + o.numel() * 4
    out['pipeline'] = dict(batch=[list(im.shape) for im in imgs],
                           out=list(o.shape), call_ms=t_call * 1e3,
                           kernel_us=t_k * 1e6, bytes=nbytes,
                           kernel_GBps=nbytes / t_k / 1e9)
    print('pipeline', out['pipeline'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', default='r01')
    ap.add_argument('--only', default='kl,conv')
    ap.add_argument('--no-miopen', action='store_true')
    args = ap.parse_args()
    out = dict(device=torch.cuda.get_device_name(0), time=time.time())
    if 'kl' in args.only.split(','):
        bench_kl(out)
    if 'conv' in args.only.split(','):
        bench_conv(out, not args.no_miopen)
    if 'tiles' in args.only.split(','):
        bench_conv_tiles(out)
    if 'stream' in args.only.split(','):
        bench_stream_sweep(out)
    if 'bf16' in args.only.split(','):
        bench_bf16(out)
    if 'pipeline' in args.only.split(','):
        bench_pipeline(out)
    if 'wgrad' in args.only.split(','):
        bench_wgrad(out)
    if 'c8' in args.only.split(','):
        bench_c8(out)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    path = os.path.join(REPO, 'gpurun_out', f'kernels_{args.tag}.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
