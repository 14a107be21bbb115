"""GPU tests (-m gpu): every HIP layer kernel, called through the C ABI,
against a plain PyTorch fp32 reference of the same op computed on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _close(got, ref, rtol=2e-4, atol_rel=2e-5, what=''):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, f'{what}: shape {got.shape} vs {ref.shape}'
    scale = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs()
    tol = atol_rel * scale + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(int(err.argmax()), tuple(err.shape))
        nbad = int(bad.sum())
        raise AssertionError(
            f'{what}: {nbad}/{bad.numel()} elements off; max err '
            f'{float(err.max()):.3e} at {idx} (got {float(got[idx]):.6f} ref '
            f'{float(ref[idx]):.6f}, scale {scale:.3e}); first bad idx '
            f'{tuple(int(v) for v in bad.nonzero()[0])}')


def test_mfma_layout_identity():
    """1x1 conv with a permutation-like weight on an asymmetric input: catches
    any row/col swap in the MFMA fragment or accumulator mapping."""
    from ld_amd import layers as Y
    dev = _dev()
    N, C, H, W = 1, 64, 8, 40
    x = (torch.arange(N * C * H * W, dtype=torch.float32).reshape(N, C, H, W)
         % 977) / 97.0
    w = torch.zeros(64, 64, 1, 1)
    perm = (torch.arange(64) * 7 + 3) % 64
    for co in range(64):
        w[co, perm[co], 0, 0] = 1.0 + co / 64.0
    y, _ = Y.conv_forward_raw(x.to(dev).reshape(N, C, -1), w.to(dev), 1, 0,
                              ((H, W), ))
    ref = F.conv2d(x, w)
    _close(y.reshape(ref.shape), ref, what='identity conv')


CONV_CASES = [
    # name, N, Cin, Cout, k, stride, pad, levels
    ('1x1_64_256', 2, 64, 256, 1, 1, 0, ((20, 34), )),
    ('1x1_s2_256_512', 2, 256, 512, 1, 2, 0, ((20, 34), )),
    ('1x1_s2_odd', 1, 64, 128, 1, 2, 0, ((13, 21), )),
    ('3x3_64_64', 2, 64, 64, 3, 1, 1, ((24, 40), )),
    ('3x3_s2_128', 2, 128, 128, 3, 2, 1, ((26, 42), )),
    ('3x3_s2_odd', 1, 64, 64, 3, 2, 1, ((13, 21), )),
    ('3x3_256_256_levels', 2, 256, 256, 3, 1, 1,
     ((20, 28), (10, 14), (5, 7), (3, 4), (2, 2))),
    ('3x3_256_80_levels', 2, 256, 80, 3, 1, 1,
     ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))),
    ('3x3_256_68_levels', 1, 256, 68, 3, 1, 1,
     ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))),
    ('1x1_2048_256', 1, 2048, 256, 1, 1, 0, ((7, 11), )),
    ('3x3_512_512_tiny', 2, 512, 512, 3, 1, 1, ((4, 6), )),
]


def _ref_conv_levels(x3, w, b, stride, pad, levels):
    outs, off = [], 0
    for h, wd in levels:
        xl = x3[:, :, off:off + h * wd].reshape(x3.shape[0], x3.shape[1], h, wd)
        outs.append(F.conv2d(xl, w, b, stride=stride, padding=pad).flatten(2))
        off += h * wd
    return torch.cat(outs, 2)


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(case):
    from ld_amd import layers as Y
    dev = _dev()
    name, N, cin, cout, k, stride, pad, levels = case
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    P = sum(h * w for h, w in levels)
    x = torch.randn(N, cin, P, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
    b = torch.randn(cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = _ref_conv_levels(xr, wr, br, stride, pad, levels)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y, out_levels = Y.conv2d(xd, wd, bd, stride, pad, levels)
    assert sum(h * w for h, w in out_levels) == ref.shape[2]
    _close(y, ref, what=name + ' fwd')
    y.backward(go.to(dev))
    _close(xd.grad, xr.grad, what=name + ' dgrad')
    _close(wd.grad, wr.grad, what=name + ' wgrad')
    _close(bd.grad, br.grad, what=name + ' bias grad')


STREAM_SHAPES = ['2x2x2x8', '2x2x1x8', '2x1x2x8', '1x2x2x8', '1x2x4x8', '1x2x1x8',
                 '1x1x2x8', '1x1x1x8', '1x1x4x8', '3x1x1x8', '3x2x1x8', '0',
                 '1x1x1x8x4', '1x2x1x8x4', '2x1x1x8x4', '2x2x1x8x4',
                 '1x1x1x16', '1x1x2x16', '1x1x4x16', '2x1x2x16', '1x2x2x16',
                 '2x1x1x16', '1x1x1x16x4', '2x1x1x16x4', '1x2x1x16x4']


@pytest.mark.parametrize('shape', STREAM_SHAPES)
def test_conv_stream_shapes(shape, monkeypatch):
    """Every register-tile shape of the streaming conv kernel (and the LDS
    kernel, '0') on every case: forward and data-gradient against F.conv2d.
    The autotuner may pick any of them, so each must be right on its own."""
    from ld_amd import layers as Y
    monkeypatch.setenv('LD_CONV_STREAM', shape)
    dev = _dev()
    for case in CONV_CASES:
        name, N, cin, cout, k, stride, pad, levels = case
        g = torch.Generator().manual_seed(len(name) * 7 + cin)
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
        xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
        ref = _ref_conv_levels(xr, wr, None, stride, pad, levels)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd, wd = (t.to(dev).requires_grad_(True) for t in (x, w))
        y, _ = Y.conv2d(xd, wd, None, stride, pad, levels)
        _close(y, ref, what=f'{name} fwd [{shape}]')
        y.backward(go.to(dev))
        _close(xd.grad, xr.grad, what=f'{name} dgrad [{shape}]')


VEC_SHAPES = ['1x4x1x8x2', '1x4x2x8x2', '1x4x4x8x2', '2x4x1x8x2', '2x4x2x8x2',
              '1x2x1x8x2', '1x2x2x8x2', '1x2x4x8x2', '2x2x2x8x2', '1x4x2x16x2',
              '1x4x4x16x2', '1x2x4x16x2']
VEC_CASES = [  # N, cin, cout, (h, w): 1x1 stride-1 convs
    (2, 256, 1024, (50, 84)), (2, 64, 256, (20, 28)), (1, 1024, 256, (25, 42)),
    (2, 512, 128, (13, 22)), (1, 128, 80, (9, 12)), (2, 32, 96, (7, 11)),
]


@pytest.mark.parametrize('shape', VEC_SHAPES)
def test_conv1x1_vector_shapes(shape, monkeypatch):
    """conv1x1_vec_kernel (vector operand loads, strided column tiles, 8/16-byte
    epilogue): forward, data gradient, and the fused BN / residual / ReLU /
    raw-output epilogue, against F.conv2d.  Cases whose position count is not a
    multiple of the vector width (25x42, 7x11) must fall back to another shape
    and still be right."""
    from ld_amd import layers as Y
    monkeypatch.setenv('LD_CONV_STREAM', shape)
    dev = _dev()
    for N, cin, cout, (h, w_) in VEC_CASES:
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(N, cin, h * w_, generator=g)
        w = torch.randn(cout, cin, 1, 1, generator=g) / cin**0.5
        xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
        ref = F.conv2d(xr.view(N, cin, h, w_), wr).reshape(N, cout, -1)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd, wd = (t.to(dev).requires_grad_(True) for t in (x, w))
        y, _ = Y.conv2d(xd, wd, None, 1, 0, ((h, w_), ))
        _close(y, ref, what=f'1x1 {cin}>{cout} fwd [{shape}]')
        y.backward(go.to(dev))
        _close(xd.grad, xr.grad, what=f'1x1 {cin}>{cout} dgrad [{shape}]')
        # fused epilogue incl. the raw second output
        gamma, beta = torch.rand(cout, generator=g) + .5, torch.randn(cout, generator=g)
        mean, var = torch.randn(cout, generator=g) * .1, torch.rand(cout, generator=g) + .5
        res = torch.randn(N, cout, h * w_, generator=g)
        gr = gamma.clone().requires_grad_(True)
        refz = F.relu(F.batch_norm(ref.detach().view(N, cout, h, w_), mean, var, gr,
                                   beta, False, 0.0, 1e-5).reshape(N, cout, -1) + res)
        refz.backward(go)
        gd = gamma.to(dev).requires_grad_(True)
        z, _ = Y.conv_bn_act(x.to(dev).requires_grad_(True), w.to(dev), gd,
                             beta.to(dev), mean.to(dev), var.to(dev), 1e-5, 1, 0,
                             ((h, w_), ), res.to(dev), True)
        _close(z, refz, what=f'1x1 {cin}>{cout} conv+bn+res+relu [{shape}]')
        z.backward(go.to(dev))  # d(gamma) reads the raw second output
        _close(gd.grad, gr.grad, rtol=1e-3, atol_rel=1e-4,
               what=f'1x1 {cin}>{cout} dgamma via y_raw [{shape}]')


@pytest.mark.parametrize('mode', ['0', '16', '32'])
def test_conv_wgrad_kernels(mode, monkeypatch):
    """Weight gradient under each kernel (LD_CONV_WGRAD: 0 = workgroup tiles,
    16/32 = wave-private tiles) against autograd of F.conv2d."""
    from ld_amd import layers as Y
    monkeypatch.setenv('LD_CONV_WGRAD', mode)
    dev = _dev()
    for case in CONV_CASES:
        name, N, cin, cout, k, stride, pad, levels = case
        g = torch.Generator().manual_seed(len(name) * 11 + cout)
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
        xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
        ref = _ref_conv_levels(xr, wr, None, stride, pad, levels)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd, wd = (t.to(dev).requires_grad_(True) for t in (x, w))
        y, _ = Y.conv2d(xd, wd, None, stride, pad, levels)
        y.backward(go.to(dev))
        _close(wd.grad, wr.grad, what=f'{name} wgrad [{mode}]')


WGRAD_TILE_CFGS = ['1,1,32,1,0', '1,1,32,3,0', '1,1,32,3,1', '1,2,32,1,0', '1,2,32,2,1',
                   '1,2,32,5,0', '1,2,32,5,1', '1,4,32,1,0', '1,4,32,3,1', '1,2,64,2,1',
                   '1,2,64,3,0', '1,4,64,1,0', '1,4,64,2,1', '1,4,64,4,0',
                   '2,0,0,1,0', '2,0,0,3,0', '2,0,0,8,0']


@pytest.mark.parametrize('cfg', WGRAD_TILE_CFGS)
def test_conv_wgrad_tile_configs(cfg, monkeypatch):
    """The workgroup-tiled fp32 weight gradient (conv_wgrad.hip) under every
    (k-groups, slice width) instance, with 1..5 workgroups per tile combined by
    the slab reduce launch and inside the launch (last-arriving workgroup),
    against autograd of F.conv2d.  Covers ragged 128-channel tiles (68 / 80 /
    64 channels), strides, the level-concatenated head maps and reductions
    shorter than one slice per split."""
    from ld_amd import layers as Y
    monkeypatch.setenv('LD_CONV_WGRAD_CFG', cfg)
    dev = _dev()
    for case in CONV_CASES:
        name, N, cin, cout, k, stride, pad, levels = case
        g = torch.Generator().manual_seed(len(name) * 13 + cout)
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
        xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
        ref = _ref_conv_levels(xr, wr, None, stride, pad, levels)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd, wd = (t.to(dev).requires_grad_(True) for t in (x, w))
        y, _ = Y.conv2d(xd, wd, None, stride, pad, levels)
        y.backward(go.to(dev))
        _close(wd.grad, wr.grad, what=f'{name} wgrad tile [{cfg}]')


def _wgrad_call(lib, d, x, dy, dw, ws, acc, st):
    import ctypes as C
    from ld_amd import lib as L
    L.check(lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), acc,
                              L.ptr(ws), ws.numel(), st), 'ld_conv_wgrad')


@pytest.mark.parametrize('shape', [(256, 256, 3, 1, 1, ((50, 84), )),
                                   (1024, 256, 1, 1, 0, ((50, 84), )),
                                   (256, 80, 3, 1, 1, ((20, 28), (10, 14), (5, 7)))])
def test_conv_wgrad_fused_equals_slab_reduce(shape, monkeypatch):
    """In-launch combination of the split partials == the fixed-order slab
    reduce launch, BIT FOR BIT, overwrite and accumulate, and stays so over
    repeated launches on one workspace with alternating operands while another
    stream keeps the device busy (the hand-off under load: a stale partial or a
    missed ticket shows as a differing word; every word is compared)."""
    import ctypes as C
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    cin, cout, k, stride, pad, levels = shape
    N = 2
    d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
    g = torch.Generator().manual_seed(cin + cout)
    xs = [torch.randn(N, cin, d.Pin, generator=g).to(dev) for _ in range(2)]
    dys = [torch.randn(N, cout, d.Pout, generator=g).to(dev) for _ in range(2)]
    # sized before the plans below are forced: the worst case over every plan
    ws = torch.zeros(lib.ld_conv_tune_wgrad_workspace_bytes(C.byref(d)),
                     dtype=torch.uint8, device=dev)
    st = L.stream_ptr(dev)
    base = torch.randn(cout, cin, k, k, generator=g).to(dev)
    for kg, bk, sp in ((2, 32, 6), (1, 32, 4), (4, 64, 3)):
        refs = []
        monkeypatch.setenv('LD_CONV_WGRAD_CFG', f'1,{kg},{bk},{sp},0')
        for x, dy in zip(xs, dys):
            dw = torch.empty(cout, cin, k, k, device=dev)
            _wgrad_call(lib, d, x, dy, dw, ws, 0, st)
            dwa = base.clone()
            _wgrad_call(lib, d, x, dy, dwa, ws, 1, st)
            refs.append((dw, dwa))
        torch.cuda.synchronize()
        monkeypatch.setenv('LD_CONV_WGRAD_CFG', f'1,{kg},{bk},{sp},1')
        side = torch.cuda.Stream()
        busy = torch.randn(4096, 4096, device=dev)
        for it in range(24):
            with torch.cuda.stream(side):  # uneven load on the other queue
                for _ in range(1 + it % 3):
                    busy = busy * 1.0001 + 0.5
            x, dy = xs[it % 2], dys[it % 2]
            dw = torch.full((cout, cin, k, k), float('nan'), device=dev)
            _wgrad_call(lib, d, x, dy, dw, ws, 0, st)
            dwa = base.clone()
            _wgrad_call(lib, d, x, dy, dwa, ws, 1, st)
            torch.cuda.synchronize()
            assert torch.equal(dw, refs[it % 2][0]), (kg, bk, sp, it, 'overwrite')
            assert torch.equal(dwa, refs[it % 2][1]), (kg, bk, sp, it, 'accumulate')


def test_conv_tune_wgrad_records_a_pick(tmp_path):
    """ld_conv_tune_wgrad times the candidates on the caller's buffers, stores
    the winner under the MODE 2 key, and the next launch uses it (result still
    equal to the wave-private kernel's within fp32 summation-order noise)."""
    import ctypes as C
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    d, _ = Y.conv_desc(2, 128, 256, 3, 3, 1, 1, ((30, 44), ))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 128, d.Pin, generator=g).to(dev)
    dy = torch.randn(2, 256, d.Pout, generator=g).to(dev)
    ws = torch.zeros(lib.ld_conv_tune_wgrad_workspace_bytes(C.byref(d)),
                     dtype=torch.uint8, device=dev)
    st = L.stream_ptr(dev)
    dw = torch.empty(256, 128, 3, 3, device=dev)
    rc = lib.ld_conv_tune_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws),
                                ws.numel(), st)
    assert rc == 0
    assert lib.ld_conv_tune_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws),
                                  ws.numel(), st) == 1
    f = str(tmp_path / 't.txt')
    L.save_tune_table(f)
    mine = [ln.split() for ln in open(f) if ln.split()[:5] == ['2', '128', '256', '3', '3']]
    assert len(mine) == 1 and len(mine[0]) == 24
    _wgrad_call(lib, d, x, dy, dw, ws, 0, st)
    ref = torch.nn.grad.conv2d_weight(
        x.reshape(2, 128, 30, 44).cpu().double(), (256, 128, 3, 3),
        dy.reshape(2, 256, 30, 44).cpu().double(), padding=1)
    _close(dw, ref, what='tuned wgrad')


@pytest.mark.parametrize('shape', [(2, 5, 40, 64), (1, 3, 37, 132), (2, 4, 9, 8),
                                   (1, 2, 16, 30), (3, 1, 7, 7), (1, 64, 400, 672)])
def test_maxpool_vector_and_scalar_paths(shape):
    """max_pool2d(3, 2, 1): the 16-byte-load kernel (W % 4 == 0: two outputs per
    thread, left column by lane shuffle) and the scalar fallback, bit-exact
    against torch on CPU (a max has no rounding)."""
    from ld_amd import layers as Y
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    got = Y.maxpool3x3s2(x.to(_dev())).cpu()
    assert torch.equal(got, F.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize('hw', [(38, 50), (64, 256), (131, 203), (800, 1344)])
def test_stem_lds_kernel_bit_identical_to_gather_kernel(hw, monkeypatch):
    """conv_stem_lds_kernel (round 6: patch + weight image through LDS) against
    conv_stem_kernel (LD_CONV_STEM=stream, per-element gathers): same k pairs in
    the same order on the same operand values -> identical bits, incl. ragged
    right / bottom tiles and the zero padding."""
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(hw[0] + hw[1])
    x = torch.randn(2, 3, hw[0], hw[1], generator=g).to(dev)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(dev)
    gamma, beta = (torch.rand(64, generator=g) + .5).to(dev), torch.randn(64, generator=g).to(dev)
    mean, var = (torch.randn(64, generator=g) * .1).to(dev), (torch.rand(64, generator=g) + .5).to(dev)
    outs = []
    for sel in ('stream', 'lds'):
        monkeypatch.setenv('LD_CONV_STEM', sel)
        y, lv = Y.conv_bn_act_infer(x.reshape(2, 3, -1), w, gamma, beta, mean, var,
                                    1e-5, 2, 3, (hw, ))
        torch.cuda.synchronize()
        outs.append(y.clone())
    assert torch.equal(outs[0], outs[1])
    if hw[0] <= 131:
        ref = F.relu(F.batch_norm(F.conv2d(x.cpu(), w.cpu(), stride=2, padding=3),
                                  mean.cpu(), var.cpu(), gamma.cpu(), beta.cpu(),
                                  False, 0.0, 1e-5))
        _close(outs[1].reshape(ref.shape), ref, what='stem (LDS kernel)')


def test_conv_fused_epilogue_and_stem():
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    # stem: 7x7 s2 p3, Cin=3, BN(eval)+ReLU folded
    x = torch.randn(2, 3, 38, 50, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    gamma, beta = torch.rand(64, generator=g) + .5, torch.randn(64, generator=g)
    mean, var = torch.randn(64, generator=g) * .1, torch.rand(64, generator=g) + .5
    ref = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=3), mean, var,
                              gamma, beta, False, 0.0, 1e-5))
    y, lv = Y.conv_bn_act_infer(x.to(dev).reshape(2, 3, -1), w.to(dev),
                                gamma.to(dev), beta.to(dev), mean.to(dev),
                                var.to(dev), 1e-5, 2, 3, ((38, 50), ))
    assert lv == ((19, 25), )
    _close(y.reshape(ref.shape), ref, what='stem conv+bn+relu')
    mp = Y.maxpool3x3s2(y.reshape(ref.shape))
    _close(mp, F.max_pool2d(ref, 3, 2, 1), what='maxpool')
    # bottleneck tail: conv1x1 -> BN -> + residual -> ReLU in one launch
    x = torch.randn(2, 64, 10, 14, generator=g)
    w = torch.randn(256, 64, 1, 1, generator=g) * 0.1
    res = torch.randn(2, 256, 10, 14, generator=g)
    gamma, beta = torch.rand(256, generator=g), torch.randn(256, generator=g)
    mean, var = torch.randn(256, generator=g) * .1, torch.rand(256, generator=g) + .5
    ref = F.relu(F.batch_norm(F.conv2d(x, w), mean, var, gamma, beta, False,
                              0.0, 1e-5) + res)
    y, _ = Y.conv_bn_act_infer(x.to(dev).reshape(2, 64, -1), w.to(dev),
                               gamma.to(dev), beta.to(dev), mean.to(dev),
                               var.to(dev), 1e-5, 1, 0, ((10, 14), ),
                               residual=res.to(dev).reshape(2, 256, -1))
    _close(y.reshape(ref.shape), ref, what='conv+bn+res+relu')


@pytest.mark.parametrize('relu,with_res', [(True, False), (True, True),
                                           (False, False)])
def test_bn_act(relu, with_res):
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    N, C, H, W = 2, 96, 13, 21
    x = torch.randn(N, C, H * W, generator=g)
    res = torch.randn(N, C, H * W, generator=g) if with_res else None
    gamma = torch.rand(C, generator=g) + .5
    beta = torch.randn(C, generator=g)
    mean, var = torch.randn(C, generator=g), torch.rand(C, generator=g) + .5
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    rr = res.clone().requires_grad_(True) if with_res else None
    ref = F.batch_norm(xr.reshape(N, C, H, W), mean, var, gr, br, False, 0.0,
                       1e-5).reshape(N, C, -1)
    if with_res:
        ref = ref + rr
    if relu:
        ref = F.relu(ref)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, gd, bd = (t.to(dev).requires_grad_(True) for t in (x, gamma, beta))
    rd = res.to(dev).requires_grad_(True) if with_res else None
    y = Y.bn_act(xd, gd, bd, mean.to(dev), var.to(dev), 1e-5, rd, relu)
    _close(y, ref, what='bn fwd')
    y.backward(go.to(dev))
    _close(xd.grad, xr.grad, what='bn dx')
    _close(gd.grad, gr.grad, what='bn dgamma')
    _close(bd.grad, br.grad, what='bn dbeta')
    if with_res:
        _close(rd.grad, rr.grad, what='bn dres')


def test_gn_act_levels():
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    levels = ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))
    N, C, G = 2, 64, 32
    P = sum(h * w for h, w in levels)
    x = torch.randn(N, C, P, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + .5, torch.randn(C, generator=g)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    outs, off = [], 0
    for h, w in levels:
        xl = xr[:, :, off:off + h * w].reshape(N, C, h, w)
        outs.append(F.relu(F.group_norm(xl, G, gr, br, 1e-5)).flatten(2))
        off += h * w
    ref = torch.cat(outs, 2)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, gd, bd = (t.to(dev).requires_grad_(True) for t in (x, gamma, beta))
    y = Y.gn_act(xd, gd, bd, G, 1e-5, levels, True)
    _close(y, ref, what='gn fwd')
    y.backward(go.to(dev))
    _close(xd.grad, xr.grad, rtol=5e-4, what='gn dx')
    _close(gd.grad, gr.grad, what='gn dgamma')
    _close(bd.grad, br.grad, what='gn dbeta')


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('cfg', [(64, 32, True), (256, 32, True), (48, 12, False),
                                 (64, 16, True)])
def test_round6_norm_backward_kernels_bit_identical(cfg, mode, monkeypatch):
    """GroupNorm backward (round 6: the row's statistics in a scalar table, every
    load issued up front) against the round-5 kernels (LD_NN_OLD=1): dx, its C8
    image in bf16 mode, dgamma, dbeta -- identical bits, on a pyramid whose small
    levels put level boundaries inside a thread's four positions, with 2, 4 and 8
    channels per group."""
    from ld_amd import layers as Y
    dev = _dev()
    C, G, relu = cfg
    levels = ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2)) if C < 256 else \
        ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))
    P = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(C + G)
    x = (torch.randn(2, C, P, generator=g) * 2 + 0.5).to(dev)
    gamma = (torch.rand(C, generator=g) + .5).to(dev)
    beta = torch.randn(C, generator=g).to(dev)
    go = torch.randn(2, C, P, generator=g).to(dev)
    Y.set_precision(mode)
    try:
        outs = []
        for old in ('1', '0'):
            monkeypatch.setenv('LD_NN_OLD', old)
            xd, gd, bd = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
            y = Y.gn_act(xd, gd, bd, G, 1e-5, levels, relu)
            y.backward(go)
            torch.cuda.synchronize()
            img = Y._c8_cached(xd.grad)
            outs.append((xd.grad.clone(), gd.grad.clone(), bd.grad.clone(),
                         None if img is None else img.clone()))
        for a, b, what in zip(outs[0], outs[1], ('dx', 'dgamma', 'dbeta', 'dx C8')):
            if a is None:
                assert b is None
                continue
            assert torch.equal(a, b), (cfg, mode, what)
    finally:
        Y.set_precision('fp32')


@pytest.mark.parametrize('cfg', [(64, 8, True), (64, 32, True), (256, 32, True),
                                 (64, 16, False)])
def test_gn_lean_backward_bit_identical(cfg):
    """Round 6, bf16 mode: the lean GroupNorm backward (ld_gn_backward_c8_lean)
    recomputes the ReLU mask from x with the forward's own expression instead of
    reading y -- dx, its C8 image, dgamma and dbeta carry the same bits as the
    y-reading kernels (LD_GN_LEAN=0), on pyramids with level boundaries inside a
    thread's four positions."""
    from ld_amd import layers as Y
    dev = _dev()
    C, G, relu = cfg
    levels = ((12, 20), (6, 10), (3, 5), (2, 3), (1, 3)) if C < 256 else \
        ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))
    P = sum(h * w for h, w in levels)
    assert P % 4 == 0
    g = torch.Generator().manual_seed(C * 3 + G)
    x = (torch.randn(2, C, P, generator=g) * 2 + 0.5).to(dev)
    gamma = (torch.randn(C, generator=g) * .7).to(dev)  # both signs
    beta = torch.randn(C, generator=g).to(dev)
    go = torch.randn(2, C, P, generator=g).to(dev)
    Y.set_precision('bf16')
    try:
        outs = []
        for lean in (False, True):
            Y._GN_LEAN[0] = lean
            xd, gd, bd = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
            y = Y.gn_act(xd, gd, bd, G, 1e-5, levels, relu)
            seen = {}
            xd.register_hook(lambda gr: seen.setdefault('dx', gr))
            y.backward(go)
            torch.cuda.synchronize()
            img = Y._c8_cached(seen['dx'])
            assert img is not None
            outs.append((y.detach().clone(), xd.grad.clone(), gd.grad.clone(),
                         bd.grad.clone(), img.clone()))
        for a, b, what in zip(outs[0], outs[1], ('y', 'dx', 'dgamma', 'dbeta', 'dx C8')):
            assert torch.equal(a, b), (cfg, what)
    finally:
        Y._GN_LEAN[0] = True
        Y.set_precision('fp32')


def test_fused_conv_gn_node_bit_identical_to_the_pair():
    """Round 6, bf16 mode: ConvGnActFn (one autograd node per trainable tower
    layer; the gradient of the conv output exists only as its C8 image) against
    conv2d + gn_act: output and every gradient identical, two stacked layers so
    that the second layer's data gradient feeds the first layer's GN backward."""
    from ld_amd import layers as Y
    dev = _dev()
    levels = ((12, 20), (6, 10), (3, 5), (2, 3), (1, 3))
    P = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(77)
    C = 64
    base = [torch.randn(2, C, P, generator=g)] + [
        t for _ in range(2) for t in (torch.randn(C, C, 3, 3, generator=g) * 0.05,
                                      torch.rand(C, generator=g) + .5,
                                      torch.randn(C, generator=g) * .3)]
    go = torch.randn(2, C, P, generator=g).to(dev)
    Y.set_precision('bf16')
    try:
        outs = []
        # third variant: the first layer hands its output on ONLY as the C8 image
        # (c8_only: the fp32 tensor is an unwritten placeholder, layers._unwritten)
        for fused, c8_only in ((False, False), (True, False), (True, True)):
            Y._FUSE_CONV_GN[0] = fused
            t = [b.to(dev).requires_grad_(True) for b in base]
            h, lv = Y.conv_gn_act(t[0], t[1], t[2], t[3], 32, 1e-5, 1, 1, levels,
                                  c8_only=c8_only)
            assert Y._unwritten(h) == c8_only
            y, _ = Y.conv_gn_act(h, t[4], t[5], t[6], 32, 1e-5, 1, 1, lv)
            y.backward(go)
            torch.cuda.synchronize()
            outs.append([y.detach().clone()] + [v.grad.clone() for v in t])
        for other in outs[1:]:
            for i, (a, b) in enumerate(zip(outs[0], other)):
                assert torch.equal(a, b), i
    finally:
        Y._FUSE_CONV_GN[0] = True
        Y.set_precision('fp32')


@pytest.mark.parametrize('fine,coarse', [((50, 84), (25, 42)),
                                         ((25, 25), (13, 13)),
                                         ((13, 21), (7, 11))])
def test_upsample_add(fine, coarse):
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    a = torch.randn(2, 8, *fine, generator=g)
    b = torch.randn(2, 8, *coarse, generator=g)
    ar, brr = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ar + F.interpolate(brr, size=fine, mode='nearest')
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = Y.upsample_add(ad, bd)
    _close(y, ref, what='upsample_add fwd')
    y.backward(go.to(dev))
    _close(ad.grad, ar.grad, what='d fine')
    _close(bd.grad, brr.grad, what='d coarse')


def test_scale_levels_and_sgd():
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(14)
    levels = ((6, 10), (3, 5), (2, 3))
    P = sum(h * w for h, w in levels)
    x = torch.randn(2, 68, P, generator=g)
    s = torch.rand(3, generator=g) + .5
    xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    sc = torch.cat([sr[i].expand(h * w) for i, (h, w) in enumerate(levels)])
    ref = xr * sc
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, sd = x.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
    y = Y.scale_levels(xd, sd, levels)
    _close(y, ref, what='scale fwd')
    y.backward(go.to(dev))
    _close(xd.grad, xr.grad, what='scale dx')
    _close(sd.grad, sr.grad, what='dscale')
    # SGD vs torch.optim.SGD, two steps, odd length (tail path)
    n = 4099
    p = torch.randn(n, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd = p.to(dev)
    buf = torch.zeros(n, device=dev)
    for _ in range(2):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone()
        opt.step()
        Y.sgd_step(pd, gr.to(dev), buf, 0.01, 0.9, 1e-4)
    _close(pd, pr, rtol=1e-5, atol_rel=1e-6, what='sgd')


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', [  # cin, cout, k, stride, (h, w), residual, relu
    (64, 64, 3, 1, (20, 28), False, True), (64, 256, 1, 1, (20, 28), True, True),
    (128, 128, 3, 2, (21, 27), False, True), (256, 512, 1, 2, (16, 12), False, False),
])
def test_fused_conv_bn_matches_unfused_pair(mode, case):
    """layers.ConvBnActFn (one forward launch: conv with the folded eval-BN,
    residual and ReLU in its epilogue + the raw conv result) against the
    round-2 pair conv2d -> bn_act on the same inputs: the same output and the
    same gradients for x, w, gamma, beta and the residual.  Both sides run the
    same conv / BN-backward kernels; only the forward affine moves into the conv
    epilogue (fp32 operation order may differ by one rounding)."""
    from ld_amd import layers as Y
    cin, cout, k, stride, (h, w_), has_res, relu = case
    dev = _dev()
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(2, cin, h * w_, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
    gamma, beta = torch.rand(cout, generator=g) + .5, torch.randn(cout, generator=g)
    mean, var = torch.randn(cout, generator=g) * .1, torch.rand(cout, generator=g) + .5
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w_ + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(2, cout, ho * wo, generator=g) if has_res else None
    go = torch.randn(2, cout, ho * wo, generator=g)
    Y.set_precision(mode)
    outs = []
    try:
        for fused in (False, True):
            Y._FUSE_CONV_BN[0] = fused
            leaves = [t.to(dev).requires_grad_(True) for t in (x, wt, gamma, beta)]
            r = res.to(dev).requires_grad_(True) if has_res else None
            z, lv = Y.conv_bn_act(leaves[0], leaves[1], leaves[2], leaves[3],
                                  mean.to(dev), var.to(dev), 1e-5, stride, k // 2,
                                  ((h, w_), ), r, relu)
            assert lv == ((ho, wo), )
            z.backward(go.to(dev))
            outs.append([z.detach().cpu()] + [t.grad.cpu() for t in leaves] +
                        ([r.grad.cpu()] if has_res else []))
    finally:
        Y._FUSE_CONV_BN[0] = True
        Y.set_precision('fp32')
    names = ['z', 'dx', 'dw', 'dgamma', 'dbeta', 'dres']
    for nm, a, b in zip(names, outs[0], outs[1]):
        sc = float(a.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 2e-5 * sc, (nm, err, sc)


@pytest.mark.parametrize('bf16', [False, True])
def test_weight_transform_tiled_equals_per_element(bf16):
    """The LDS-tiled batch transform (round 5) writes the same images bit for bit as
    the per-element kernels, pad rows included, on ragged channel counts."""
    import ctypes as C
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(12)
    jobs, blocks, pairs, keep = [], [], [], []
    for cout, cin, k in ((256, 256, 3), (1024, 256, 1), (64, 256, 1), (80, 256, 3),
                         (68, 256, 3), (48, 16, 3), (40, 24, 1), (512, 2048, 1)):
        w = torch.randn(cout, cin, k, k, generator=g).to(dev)
        if bf16:
            nf = lib.ld_conv_bf16_weight_image_elems(cout, cin, k, k, 0)
            nb_ = lib.ld_conv_bf16_weight_image_elems(cout, cin, k, k, 1)
            dt = torch.bfloat16
        else:
            nf = lib.ld_conv_weight_image_floats(cout, cin, k, k, 0)
            nb_ = lib.ld_conv_weight_image_floats(cout, cin, k, k, 1)
            dt = torch.float32
        ref_f = torch.empty(nf, dtype=dt, device=dev)
        ref_b = torch.empty(nb_, dtype=dt, device=dev)
        fn = lib.ld_conv_bf16_weight_transform if bf16 else lib.ld_conv_weight_transform
        L.check(fn(L.ptr(w), cout, cin, k, k, L.ptr(ref_f), L.ptr(ref_b), st), 'ref')
        out_f = torch.full((nf, ), 7.0, dtype=dt, device=dev)
        out_b = torch.full((nb_, ), 7.0, dtype=dt, device=dev)
        j = L.WtJobT()
        j.w, j.wt_fwd, j.wt_bwd = w.data_ptr(), out_f.data_ptr(), out_b.data_ptr()
        j.Cout, j.Cin, j.ntaps = cout, cin, k * k
        t = lib.ld_conv_weight_transform_tiles(cout, cin, k * k, 1 if bf16 else 0)
        assert t > 0
        jobs.append(j)
        blocks.append(t)
        pairs.append((ref_f, out_f, ref_b, out_b, (cout, cin, k)))
        keep.append(w)
    assert lib.ld_conv_weight_transform_tiles(64, 3, 49, 0) == 0  # the 7x7 stem
    tab, bmap, nb = Y._job_table(jobs, blocks, dev)
    L.check(lib.ld_conv_weight_transform_batch_tiled(L.ptr(tab), L.ptr(bmap), nb,
                                                     1 if bf16 else 0, st), 'tiled')
    torch.cuda.synchronize()
    for ref_f, out_f, ref_b, out_b, shape in pairs:
        a, b = (ref_f.view(torch.int16), out_f.view(torch.int16)) if bf16 else (ref_f, out_f)
        assert torch.equal(a, b), ('fwd', shape)
        a, b = (ref_b.view(torch.int16), out_b.view(torch.int16)) if bf16 else (ref_b, out_b)
        assert torch.equal(a, b), ('bwd', shape)
