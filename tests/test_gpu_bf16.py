"""GPU tests (-m gpu) of the bf16 compute path (BASELINE.json config 3).

Two kinds of check, with different tolerances on purpose:

 * KERNEL EXACTNESS (tight): conv_bf16.hip rounds its two matrix operands to
   bf16 (RNE) and then computes exact products with fp32 accumulation.  Against
   a plain PyTorch fp32 conv ON THE SAME ROUNDED OPERANDS the result may differ
   only by fp32 summation order: rtol 2e-4 of the element + 2e-5 of the tensor
   scale, the same bound the fp32 kernels are held to in test_gpu_layers.py.
   Every register-tile shape is forced in turn.
 * END TO END (loose, stated below): the whole C2 train step in bf16 mode
   against the reference's fp32 golden loss table.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_layers import CONV_CASES, _close, _dev, _ref_conv_levels

pytestmark = pytest.mark.gpu


def _bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.fixture
def bf16_mode():
    from ld_amd import layers as Y
    Y.set_precision('bf16')
    yield
    Y.set_precision('fp32')


BF16_CASES = [c for c in CONV_CASES if c[2] % 16 == 0] + [
    ('3x3_80_256_levels_dgradlike', 2, 80, 256, 3, 1, 1,
     ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))),
    ('1x1_1024_2048_s2', 1, 1024, 2048, 1, 2, 0, ((10, 14), )),
    # even P: the workgroup-tiled wgrad (8-byte dY pairs), incl. ragged tiles
    ('3x3_256_256_even_levels', 2, 256, 256, 3, 1, 1,
     ((20, 28), (10, 14), (6, 8))),
    ('3x3_s2_128_even', 2, 128, 128, 3, 2, 1, ((24, 40), )),
    ('1x1_256_1024_even', 1, 256, 1024, 1, 1, 0, ((10, 14), )),
    ('3x3_160_192_ragged', 1, 160, 192, 3, 1, 1, ((12, 18), )),
]


def test_bf16_layout_identity(bf16_mode):
    """Permutation-like 1x1 weights on an asymmetric, bf16-exact input: any
    row/col swap or k-order slip in the 32x32x16 fragments shows up."""
    from ld_amd import layers as Y
    dev = _dev()
    N, C, H, W = 1, 64, 8, 40
    x = ((torch.arange(N * C * H * W, dtype=torch.float32)
          .reshape(N, C, H, W) * 37) % 251) / 4.0  # <= 8 significant bits
    w = torch.zeros(96, 64, 1, 1)
    perm = (torch.arange(96) * 7 + 3) % 64
    for co in range(96):
        w[co, perm[co], 0, 0] = 1.0 + (co % 32) / 32.0
    y, _ = Y.conv_forward_raw(x.to(dev).reshape(N, C, -1), w.to(dev), 1, 0,
                              ((H, W), ))
    ref = F.conv2d(x, w)
    assert torch.equal(y.reshape(ref.shape).cpu(), ref)


@pytest.mark.parametrize('wgrad', ['wave', 'tile', 'c8tile'])
@pytest.mark.parametrize('case', BF16_CASES, ids=[c[0] for c in BF16_CASES])
def test_bf16_conv_fwd_bwd(case, wgrad, bf16_mode, monkeypatch):
    from ld_amd import layers as Y
    # both fp32-operand weight-gradient kernels (the workgroup-tiled one is
    # opt-in) with the wave-private C8 kernel where the operands are C8 images;
    # 'c8tile': the workgroup-tiled C8 kernel on EVERY case it accepts (ragged
    # 128-channel tiles included), not only where the dispatch rule picks it
    monkeypatch.setenv('LD_CONV_BF16_WGRAD', 'wave' if wgrad == 'c8tile' else wgrad)
    monkeypatch.setenv('LD_CONV_WGRAD_C8_KERNEL',
                       'tile' if wgrad == 'c8tile' else 'wave')
    dev = _dev()
    name, N, cin, cout, k, stride, pad, levels = case
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    P = sum(h * w for h, w in levels)
    x = torch.randn(N, cin, P, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
    b = torch.randn(cout, generator=g)
    # forward reference on the rounded operands
    ref = _ref_conv_levels(_bf16r(x), _bf16r(w), b, stride, pad, levels)
    go = torch.randn(ref.shape, generator=g)
    # backward references: dgrad rounds (dy, w), wgrad rounds (dy, x)
    xr = x.clone().requires_grad_(True)
    _ref_conv_levels(xr, _bf16r(w), None, stride, pad, levels).backward(
        _bf16r(go))
    wr = w.clone().requires_grad_(True)
    _ref_conv_levels(_bf16r(x), wr, None, stride, pad, levels).backward(
        _bf16r(go))
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y, _ = Y.conv2d(xd, wd, bd, stride, pad, levels)
    _close(y, ref, what=name + ' fwd')
    y.backward(go.to(dev))
    if cout % 16 == 0 or Y._PAD_DGRAD[0]:
        # (round 6: a reduction that is not a multiple of 16 -- 68 output
        # channels -- runs the bf16 kernel on a zero-padded dY)
        _close(xd.grad, xr.grad, what=name + ' dgrad')
    else:  # the fp32 kernel ran (reduction not a multiple of 16): exact operands
        xe = x.clone().requires_grad_(True)
        _ref_conv_levels(xe, w, None, stride, pad, levels).backward(go)
        _close(xd.grad, xe.grad, what=name + ' dgrad (fp32 fallback)')
    _close(wd.grad, wr.grad, what=name + ' wgrad')
    _close(bd.grad, go.sum((0, 2)), what=name + ' bias grad')


BF16_SHAPES = ['2x2x2x2x1', '2x2x1x2x1', '2x1x2x4x1', '1x2x2x2x1', '1x1x2x4x1',
               '1x1x1x4x1', '1x1x4x4x1', '2x1x4x4x1', '1x1x1x4x4', '2x1x1x4x4',
               '1x2x1x2x4', '2x2x1x2x4', '2x2x2x1x1', '1x1x2x1x1', '1x1x1x1x4',
               # LDS-tiled kernel (wvm = 0): BM/32 x BN/32 x 0 x BK x 1
               '4x4x0x32x2', '4x4x0x32x4', '2x4x0x32x4', '4x2x0x32x4',
               '4x2x0x32x2']


@pytest.mark.parametrize('shape', BF16_SHAPES)
def test_bf16_stream_shapes(shape, monkeypatch, bf16_mode):
    """Every register-tile shape of conv_stream_bf16_kernel on every case it
    fits (the tuning table may pick any of them): forward with the full fused
    epilogue and the data gradient, stride 1 and the stride-2 parity classes."""
    from ld_amd import layers as Y
    monkeypatch.setenv('LD_CONV_BF16_SHAPE', shape)
    dev = _dev()
    d = int(shape.split('x')[3])
    tiled = shape.split('x')[2] == '0'
    ran = 0
    for case in BF16_CASES:
        name, N, cin, cout, k, stride, pad, levels = case
        if tiled:
            if cin % 32 or cout % 32:
                continue
        elif (cin // 16) % d or cout % 16 or (cout // 16) % d:
            continue
        ran += 1
        g = torch.Generator().manual_seed(len(name) * 7 + cin)
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g)
        ref = _ref_conv_levels(_bf16r(x), _bf16r(w), None, stride, pad, levels)
        res = torch.randn(ref.shape, generator=g)
        ref = torch.relu(ref * scale[None, :, None] + shift[None, :, None] +
                         res)
        y, _ = Y.conv_forward_raw(x.to(dev), w.to(dev), stride, pad, levels,
                                  scale=scale.to(dev), shift=shift.to(dev),
                                  residual=res.to(dev), relu=True)
        _close(y, ref, what=f'{shape} {name} fwd+epilogue')
        go = torch.randn(ref.shape, generator=g)
        xr = x.clone().requires_grad_(True)
        _ref_conv_levels(xr, _bf16r(w), None, stride, pad, levels).backward(
            _bf16r(go))
        xd = x.to(dev).requires_grad_(True)
        yd, _ = Y.conv2d(xd, w.to(dev), None, stride, pad, levels)
        yd.backward(go.to(dev))
        _close(xd.grad, xr.grad, what=f'{shape} {name} dgrad')
    assert ran >= 1, f'no case exercises {shape}'


def test_bf16_train_step_vs_fp32_golden(golden, bf16_mode):
    """The C2 step (R50 <- R101, 2 x 800x1344) with bf16 matrix operands
    against the REFERENCE's fp32 loss table.

    Stated tolerance.  Every conv input is rounded to 8 significant bits
    (relative error <= 2^-9 per operand); the rounding errors are independent
    across the K = 576 ... 4608 terms of a dot product, so a layer perturbs its
    output by ~2^-9 relative and the ~60 student / ~110 teacher layers
    accumulate to a few 10^-3 in the logits.  Target assignment does not depend
    on the logits, so labels / positive sets are the fp32 ones bit for bit.
    Measured on an MI355X (profiles/r02_pytest_gpu_s1_bf16_rccl.txt): every
    entry of the (8 keys x 5 levels) loss table within 0.41 % of the fp32
    reference, per-parameter gradient norms median 0.65 % / p90 0.87 % / max
    1.2 % off.  Asserted at ~2.5x that (round 6, VERDICT r5: the 5x margin hid
    regressions): loss table rtol 1 % (+ atol 1e-3 for the near-empty coarse
    levels), gradient norms p90 < 2 %, max < 3 %."""
    from test_gpu_e2e import LOSS_KEYS, _setup
    name = 'c2_r50'
    g, det, batch, dbatch = _setup(golden, name, 50, 2.0)
    losses = det(**dbatch)
    table = torch.stack([torch.stack(losses[k]) for k in LOSS_KEYS])
    loss, log_vars = det._parse_losses(losses)
    loss.backward()
    torch.cuda.synchronize()
    got = table.detach().cpu().numpy().astype(np.float64)
    ref = g[name + '_losses']
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    print('bf16 loss table\n', got, '\nfp32 reference\n', ref,
          '\nrelative error per (key, level)\n', rel)
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-3)
    # targets do not depend on the precision mode (bit-exactness of labels /
    # positive sets is test_gpu_lossblock.py's job): loss_kd_neg stays exactly 0
    assert not got[LOSS_KEYS.index('loss_kd_neg')].any()
    names = [str(k) for k in g[name + '_grad_names']]
    norms = g[name + '_grad_norms']
    params = dict(det.named_parameters())
    rels = []
    for k, r in zip(names, norms):
        got_n = float(params[k].grad.double().norm())
        rels.append(abs(got_n - r) / max(r, 1e-6))
    rels = np.array(rels)
    print('grad-norm relative error: median %.3e  p90 %.3e  max %.3e (%s)' %
          (np.median(rels), np.quantile(rels, 0.9), rels.max(),
           names[int(rels.argmax())]))
    assert np.quantile(rels, 0.9) < 2e-2
    assert rels.max() < 3e-2
    assert np.isfinite(float(loss))


C8_SHAPES = ['4x4x2', '4x4x4', '2x4x4', '4x2x4', '2x2x4', '2x4x2', '4x2x2',
             '2x2x2', '4x8x2', '2x8x2',
             # 64-deep k-steps (BM/32 x BN/32 x NST x BK)
             '4x4x2x64', '2x4x2x64', '4x2x2x64', '2x2x2x64', '2x2x4x64',
             '2x4x4x64',
             # LDS image written after the barrier (... x BK x SCH)
             '4x4x4x32x1', '2x4x4x32x1', '4x2x4x32x1', '2x2x4x32x1',
             '2x2x4x64x1', '2x4x4x64x1',
             # the 8-wave LDS-DMA kernel of conv_t256.hip (NST field = 8)
             '8x8x8x64', '8x6x8x64', '4x8x8x64', '8x4x8x64']


def test_to_c8_layout():
    """ld_conv_to_c8: (N, C, P) fp32 -> (N, C/8, P, 8) bf16, RNE."""
    from ld_amd import layers as Y
    dev = _dev()
    x = torch.randn(2, 48, 77, device=dev) * 3
    got = Y.to_c8(x).reshape(2, 6, 77, 8)
    want = x.to(torch.bfloat16).reshape(2, 6, 8, 77).permute(0, 1, 3, 2)
    assert torch.equal(got, want)
    assert Y.to_c8(x) is Y.to_c8(x)  # cached on the tensor
    x.add_(1.0)                       # ... until it changes
    want = x.to(torch.bfloat16).reshape(2, 6, 8, 77).permute(0, 1, 3, 2)
    assert torch.equal(Y.to_c8(x).reshape(2, 6, 77, 8), want)


@pytest.mark.parametrize('shape', C8_SHAPES)
def test_c8_kernel_bit_identical_to_fp32_input_tile_kernel(shape, monkeypatch,
                                                           bf16_mode):
    """The C8-input kernel feeds the SAME bf16 operands to the same MFMA
    sequence as the LDS-tiled kernel that reads fp32 activations: forward with
    the fused epilogue and both data-gradient modes agree bit for bit, for
    every tile shape."""
    from ld_amd import layers as Y
    dev = _dev()
    ran = 0
    for case in BF16_CASES:
        name, N, cin, cout, k, stride, pad, levels = case
        if cin % 32 or cout % 32:
            continue
        bk = int((shape.split('x') + ['32'])[3])
        if cin % bk or cout % bk:
            continue  # forward needs Cin % BK == 0, the data gradient Cout
        ran += 1
        g = torch.Generator().manual_seed(len(name) * 3 + cout)
        P = sum(h * w for h, w in levels)
        x = torch.randn(N, cin, P, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5
             ).to(dev)
        scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
        shift = torch.randn(cout, generator=g).to(dev)
        outs = []
        monkeypatch.setattr(Y, '_C8_ALL', True)  # every layer, not only the
        for c8 in (False, True):                  # ones the policy picks
            Y.set_c8(c8)
            monkeypatch.setenv('LD_CONV_BF16_SHAPE', '4x4x0x32x2')
            monkeypatch.setenv('LD_CONV_C8_SHAPE', shape)
            try:
                y, _ = Y.conv_forward_raw(x, w, stride, pad, levels,
                                          scale=scale, shift=shift, relu=True)
                xd = x.clone().requires_grad_(True)
                yd, _ = Y.conv2d(xd, w, None, stride, pad, levels)
                go = torch.Generator().manual_seed(7)
                yd.backward(torch.randn(yd.shape, generator=go).to(dev))
            finally:
                Y.set_c8(True)
            outs.append((y, xd.grad))
        assert torch.equal(outs[0][0], outs[1][0]), f'{name} fwd [{shape}]'
        assert torch.equal(outs[0][1], outs[1][1]), f'{name} dgrad [{shape}]'
    assert ran >= 4


@pytest.mark.parametrize('cout', [68, 80, 160])
def test_t256_kernel_ragged_cout_bit_identical(cout, monkeypatch, bf16_mode):
    """The head's output convs (256 -> 80 classes / 68 = 4 x 17 bins, gfl_head.py:
    130-133) run the 8-wave LDS-DMA kernel with a PARTLY EMPTY 128-row weight tile:
    rows past Cout are out-of-range DMA lanes (zeros) and masked in both epilogues.
    Against the 4-wave tile kernel, forward with bias, every level of a pyramid."""
    from ld_amd import layers as Y
    dev = _dev()
    monkeypatch.setattr(Y, '_C8_ALL', True)
    Y.set_c8(True)
    levels = ((20, 28), (10, 14), (6, 8), (3, 4), (2, 2))
    P = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(cout)
    x = torch.randn(2, 256, P, generator=g).to(dev)
    w = (torch.randn(cout, 256, 3, 3, generator=g) / 48.0).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    Y.to_c8(x)  # the operand image rides on the fp32 tensor, as in the step
    outs = []
    for shape in ('4x4x2', '4x8x8x64', '8x6x8x64'):
        monkeypatch.setenv('LD_CONV_C8_SHAPE', shape)
        y, _ = Y.conv_forward_raw(x, w, 1, 1, levels, bias=b)
        torch.cuda.synchronize()
        outs.append(y.clone())
    assert torch.equal(outs[0], outs[1]), 'BM = 128 tile'
    assert torch.equal(outs[0], outs[2]), 'BM = 256 tile (refused below 224 rows: the default kernel)'
    ref = _ref_conv_levels(_bf16r(x.cpu()), _bf16r(w.cpu()), b.cpu(), 1, 1, levels)
    _close(outs[1], ref.reshape(outs[1].shape), what='ragged Cout vs rounded-operand reference')


def test_gn_c8_side_output(bf16_mode):
    """GroupNorm(+ReLU) forward and backward write the C8 image of their fp32
    output in the same launch: the fp32 tensors equal the plain kernels' bit for
    bit, the image equals a from-scratch conversion, and the conv that follows
    finds it cached (no conversion launch)."""
    from ld_amd import layers as Y
    dev = _dev()
    levels = ((12, 20), (6, 10), (3, 4))
    P = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, P, generator=g).to(dev)
    gamma = (torch.rand(64, generator=g) + 0.5).to(dev)
    beta = torch.randn(64, generator=g).to(dev)
    go = torch.randn(2, 64, P, generator=g).to(dev)
    outs = []
    for c8 in (False, True):
        Y.set_c8(c8)
        try:
            xr = x.clone().requires_grad_(True)
            gr, br = gamma.clone().requires_grad_(True), \
                beta.clone().requires_grad_(True)
            y = Y.gn_act(xr, gr, br, 32, 1e-5, levels, relu=True)
            img = Y._c8_cached(y)
            if c8:
                assert img is not None
                want = y.detach().to(torch.bfloat16).reshape(
                    2, 8, 8, P).permute(0, 1, 3, 2).reshape(-1)
                assert torch.equal(img, want)
                before = dict(Y.C8_STATS)
                w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
                Y.conv_forward_raw(y.detach(), w, 1, 1, levels)  # other object
                Y.conv_forward_raw(y, w, 1, 1, levels)
                assert Y.C8_STATS['reused'] == before['reused'] + 1
            else:
                assert img is None
            y.backward(go)
            outs.append((y.detach(), xr.grad, gr.grad, br.grad))
        finally:
            Y.set_c8(True)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_conv_epilogue_c8_output(bf16_mode, monkeypatch):
    """emit_c8: the bf16 forward kernels (streaming, LDS-tiled, C8-input) write
    the C8 image of y from the epilogue; it equals a conversion of y, y itself
    is unchanged, and a chain conv -> conv runs without a conversion launch."""
    from ld_amd import layers as Y
    dev = _dev()
    monkeypatch.setattr(Y, '_C8_ALL', True)
    g = torch.Generator().manual_seed(11)
    levels = ((12, 20), (6, 10))
    P = sum(h * w for h, w in levels)
    x = torch.randn(2, 64, P, generator=g).to(dev)
    w1 = (torch.randn(96, 64, 3, 3, generator=g) * 0.05).to(dev)
    w2 = (torch.randn(64, 96, 1, 1, generator=g) * 0.1).to(dev)
    scale = (torch.rand(96, generator=g) + 0.5).to(dev)
    shift = torch.randn(96, generator=g).to(dev)
    res = torch.randn(2, 96, P, generator=g).to(dev)
    for env, val in (('LD_CONV_BF16_SHAPE', '1x1x2x4x1'),
                     ('LD_CONV_BF16_SHAPE', '2x1x1x4x4'),
                     ('LD_CONV_BF16_SHAPE', '4x2x0x32x2'),
                     ('LD_CONV_C8_SHAPE', '2x2x2')):
        monkeypatch.setenv(env, val)
        Y.set_c8(env == 'LD_CONV_C8_SHAPE')
        try:
            ref, _ = Y.conv_forward_raw(x, w1, 1, 1, levels, scale=scale,
                                        shift=shift, residual=res, relu=True)
            Y.set_c8(True)
            if env != 'LD_CONV_C8_SHAPE':
                monkeypatch.setattr(Y, '_use_c8', lambda *a, **k: False)
            y, _ = Y.conv_forward_raw(x, w1, 1, 1, levels, scale=scale,
                                      shift=shift, residual=res, relu=True,
                                      emit_c8=True)
            monkeypatch.undo()
            monkeypatch.setattr(Y, '_C8_ALL', True)
        finally:
            Y.set_c8(True)
        assert torch.equal(y, ref), val
        img = Y._c8_cached(y)
        assert img is not None, val
        want = y.to(torch.bfloat16).reshape(2, 12, 8, P).permute(
            0, 1, 3, 2).reshape(-1)
        assert torch.equal(img, want), val
        before = dict(Y.C8_STATS)
        Y.conv_forward_raw(y, w2, 1, 0, levels)
        assert Y.C8_STATS['converted'] == before['converted'], val
        assert Y.C8_STATS['reused'] == before['reused'] + 1, val


def test_conv_c8_only_output_and_c8_residual(bf16_mode, monkeypatch):
    """C8-only frozen chains: y == NULL + residual_c8.  The image written equals
    the C8 conversion of the ordinary fp32 launch fed the bf16-ROUNDED residual
    (bf16 -> fp32 is exact, so the epilogue arithmetic is the same), for every
    bf16 forward kernel; a C8Act input runs the C8-operand kernel."""
    from ld_amd import layers as Y
    dev = _dev()
    monkeypatch.setattr(Y, '_C8_ALL', True)
    g = torch.Generator().manual_seed(12)
    levels = ((12, 20), (6, 10))
    P = sum(h * w for h, w in levels)
    x = torch.randn(2, 64, P, generator=g).to(dev)
    w1 = (torch.randn(96, 64, 3, 3, generator=g) * 0.05).to(dev)
    scale = (torch.rand(96, generator=g) + 0.5).to(dev)
    shift = torch.randn(96, generator=g).to(dev)
    res = torch.randn(2, 96, P, generator=g).to(dev)
    res8 = Y.C8Act(Y.to_c8(res), res.shape)
    assert torch.equal(res8.float(), _bf16r(res))
    x8 = Y.C8Act(Y.to_c8(x), x.shape)
    for env, val, xin in (('LD_CONV_BF16_SHAPE', '1x1x2x4x1', x),
                          ('LD_CONV_BF16_SHAPE', '2x1x1x4x4', x),
                          ('LD_CONV_BF16_SHAPE', '4x2x0x32x2', x),
                          ('LD_CONV_C8_SHAPE', '2x2x2', x),
                          ('LD_CONV_C8_SHAPE', '4x4x2', x8),
                          ('LD_CONV_C8_SHAPE', '4x8x8x64', x8)):
        monkeypatch.setenv(env, val)
        if env != 'LD_CONV_C8_SHAPE':
            monkeypatch.setattr(Y, '_use_c8', lambda *a, **k: False)
        ref, _ = Y.conv_forward_raw(x, w1, 1, 1, levels, scale=scale,
                                    shift=shift, residual=_bf16r(res),
                                    relu=True)
        got, lv = Y.conv_forward_raw(xin, w1, 1, 1, levels, scale=scale,
                                     shift=shift, residual=res8, relu=True,
                                     c8_only=True)
        monkeypatch.undo()
        monkeypatch.setattr(Y, '_C8_ALL', True)
        assert isinstance(got, Y.C8Act) and got.shape == (2, 96, P), val
        assert lv == levels
        want = ref.to(torch.bfloat16).reshape(2, 12, 8, P).permute(
            0, 1, 3, 2).reshape(-1)
        assert torch.equal(got.buf, want), (env, val)
    # only the position axes of a C8Act can be regrouped
    v = got.view(2, 96, 12 * 20 + 60, 1)
    assert v.buf is got.buf and v.reshape(2, 96, -1).shape == (2, 96, P)
    with pytest.raises(Exception):
        got.reshape(2, 48, -1)


def test_c8_only_needs_bf16_mode():
    """Outside bf16 mode a C8Act operand is refused, not silently converted."""
    from ld_amd import layers as Y
    from ld_amd.lib import LdError
    dev = _dev()
    x = torch.randn(1, 64, 40, device=dev)
    w = torch.randn(64, 64, 1, 1, device=dev)
    with pytest.raises(LdError):
        Y.conv_forward_raw(x, w, 1, 0, ((5, 8), ), c8_only=True)


def test_teacher_trunk_c8_only_close_to_fp32_master(bf16_mode):
    """ResNet-50 + FPN under no_grad with ``c8_activations``: the stage outputs
    are C8Acts, the FPN outputs are ordinary fp32 tensors, and they differ from
    the fp32-master-activation bf16 run only by the bf16 rounding of the
    residual trunk (16 blocks): <= 2% of each level's scale (stated)."""
    from ld_amd import build_backbone, build_neck
    from ld_amd import layers as Y
    dev = _dev()
    torch.manual_seed(3)
    bb = build_backbone(dict(
        type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
        frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True),
        norm_eval=True, style='pytorch')).to(dev).eval()
    for m in bb.modules():  # non-trivial BN statistics
        if hasattr(m, 'running_var'):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
            m.weight.data.uniform_(0.5, 1.0)
    neck = build_neck(dict(type='FPN', in_channels=[256, 512, 1024, 2048],
                           out_channels=256, start_level=1,
                           add_extra_convs='on_output', num_outs=5)).to(
                               dev).eval()
    img = torch.randn(2, 3, 128, 192, device=dev)
    with torch.no_grad():
        feats = bb(img)
        assert all(isinstance(f, torch.Tensor) for f in feats)
        want = neck(feats)
        bb.c8_activations = True
        feats8 = bb(img)
        assert all(isinstance(f, Y.C8Act) for f in feats8)
        assert [f.shape for f in feats8] == [tuple(f.shape) for f in feats]
        got = neck(feats8)
        for f8, f in zip(feats8, feats):
            err = (f8.float() - f).abs().max().item()
            assert err <= 2e-2 * f.abs().max().item() + 1e-6, err
    for a, b in zip(got, want):
        assert isinstance(a, torch.Tensor) and a.dtype == torch.float32
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    # with gradients enabled the trunk keeps its fp32 activations
    img.requires_grad_(False)
    assert all(isinstance(f, torch.Tensor) for f in bb.train()(img))


def test_bn_act_c8_side_output(bf16_mode):
    """Eval-BN + residual + ReLU forward: fp32 y identical with and without the
    C8 side output; the image equals a conversion of y."""
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    N, c, P = 2, 64, 13 * 20
    x = torch.randn(N, c, P, generator=g).to(dev)
    res = torch.randn(N, c, P, generator=g).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    beta = torch.randn(c, generator=g).to(dev)
    mean = torch.randn(c, generator=g).to(dev)
    var = (torch.rand(c, generator=g) + 0.5).to(dev)
    outs = []
    for c8 in (False, True):
        Y.set_c8(c8)
        try:
            y = Y.bn_act(x, gamma, beta, mean, var, 1e-5, residual=res,
                         relu=True)
        finally:
            Y.set_c8(True)
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    assert Y._c8_cached(outs[0]) is None
    img = Y._c8_cached(outs[1])
    want = outs[1].to(torch.bfloat16).reshape(N, c // 8, 8, P).permute(
        0, 1, 3, 2).reshape(-1)
    assert torch.equal(img, want)


@pytest.mark.parametrize('relu,has_res', [(True, True), (False, False)])
def test_bn_act_backward_c8_side_output(bf16_mode, relu, has_res):
    """Eval-BN (+ residual) + ReLU BACKWARD with the C8 image of dx as a side
    output (ld_bn_act_backward_c8, round 3): dx and dres bit-identical to the
    plain kernel, the image equal to a conversion of dx, d(gamma) / d(beta)
    equal up to the summation order (per-block fp64 partials either way)."""
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    N, c, P = 2, 64, 21 * 28
    base = [torch.randn(N, c, P, generator=g), torch.rand(c, generator=g) + 0.5,
            torch.randn(c, generator=g), torch.randn(N, c, P, generator=g)]
    mean = (torch.randn(c, generator=g) * 0.1).to(dev)
    var = (torch.rand(c, generator=g) + 0.5).to(dev)
    go = torch.randn(N, c, P, generator=g).to(dev)
    outs = []
    for on in (False, True):
        Y._BN_BWD_C8[0] = on
        try:
            x, gamma, beta, res = (t.to(dev).requires_grad_(True) for t in base)
            y = Y.bn_act(x, gamma, beta, mean, var, 1e-5,
                         residual=res if has_res else None, relu=relu)
            seen = {}
            x.register_hook(lambda gr: seen.setdefault('dx', gr))
            y.backward(go)
        finally:
            Y._BN_BWD_C8[0] = True
        outs.append((x.grad, gamma.grad, beta.grad,
                     res.grad if has_res else None, seen['dx']))
    a, b = outs
    assert torch.equal(a[0], b[0])
    if has_res:
        assert torch.equal(a[3], b[3])
    for i in (1, 2):
        sc = float(a[i].abs().max()) + 1e-12
        assert float((a[i] - b[i]).abs().max()) <= 1e-6 * sc
    assert Y._c8_cached(a[4]) is None
    img = Y._c8_cached(b[4])
    assert img is not None, 'no C8 image attached to dx'
    want = b[0].to(torch.bfloat16).reshape(N, c // 8, 8, P).permute(
        0, 1, 3, 2).reshape(-1)
    assert torch.equal(img, want)


@pytest.mark.parametrize('has_res', [True, False])
def test_conv_bn_backward_without_the_unread_fp32_gradient(bf16_mode, has_res):
    """Round 6: in ConvBnActFn.backward (bf16 mode, C8 operands in both conv
    gradients) nothing reads the fp32 copy of d(conv output) -- the BN backward
    then writes only its C8 image (LD_DRAW_C8_ONLY, default on).  Every gradient
    bit-identical to the path that also writes the fp32 copy."""
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    N, cin, cout, H, W = 2, 64, 128, 20, 28
    lv = ((H, W), )
    base = [torch.randn(N, cin, H * W, generator=g),
            torch.randn(cout, cin, 3, 3, generator=g) * 0.05,
            torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g),
            torch.randn(N, cout, H * W, generator=g)]
    mean = (torch.randn(cout, generator=g) * 0.1).to(dev)
    var = (torch.rand(cout, generator=g) + 0.5).to(dev)
    go = torch.randn(N, cout, H * W, generator=g).to(dev)
    outs = []
    for on in (False, True):
        Y._DRAW_C8_ONLY[0] = on
        try:
            x, w, gamma, beta, res = (t.to(dev).requires_grad_(True) for t in base)
            z, _ = Y.conv_bn_act(x, w, gamma, beta, mean, var, 1e-5, 1, 1, lv,
                                 residual=res if has_res else None, relu=True)
            z.backward(go)
            torch.cuda.synchronize()
        finally:
            Y._DRAW_C8_ONLY[0] = True
        outs.append((x.grad, w.grad, gamma.grad, beta.grad,
                     res.grad if has_res else None))
    for a, b in zip(*outs):
        if a is not None:
            assert torch.equal(a, b)


@pytest.mark.parametrize('hw', [(20, 28), (25, 42)], ids=['p560', 'p1050'])
@pytest.mark.parametrize('has_res,relu', [(True, True), (False, True), (False, False)])
def test_conv_bn_lean_form_same_gradients(bf16_mode, has_res, relu, hw):
    """Round 6, ConvBnActFn in bf16 mode with a C8-operand conv: the lean form
    keeps the conv result before the affine as a bf16 C8 image
    (ld_conv_epilogue_t.y_raw_c8) and takes the ReLU mask from the C8 image of z
    (ld_bn_act_backward_c8in).  Against the fp32 form (LD_BN_LEAN=0): z, dx, dw,
    d(residual) and d(beta) carry the same bits (same mask, same sums); d(gamma)
    sees the bf16-rounded conv result -- 2^-9 relative per term; with this test's
    random-sign gradient the sum is itself of the order sqrt(n) terms, so the
    relative difference stays at that 2^-9 level (measured 2.5e-3 of the largest
    entry), the rounding every bf16 conv operand already carries."""
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    N, cin, cout, (H, W) = 2, 64, 128, hw  # 25 x 42: P % 4 == 2, two positions / thread
    lv = ((H, W), )
    base = [torch.randn(N, cin, H * W, generator=g),
            torch.randn(cout, cin, 3, 3, generator=g) * 0.05,
            torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g),
            torch.randn(N, cout, H * W, generator=g)]
    mean = (torch.randn(cout, generator=g) * 0.1).to(dev)
    var = (torch.rand(cout, generator=g) + 0.5).to(dev)
    go = torch.randn(N, cout, H * W, generator=g).to(dev)
    outs = []
    for lean in (False, True):
        Y._BN_LEAN[0] = lean
        try:
            x, w, gamma, beta, res = (t.to(dev).requires_grad_(True) for t in base)
            Y.to_c8(x)  # an image on the input: the C8-operand conv is taken
            z, _ = Y.conv_bn_act(x, w, gamma, beta, mean, var, 1e-5, 1, 1, lv,
                                 residual=res if has_res else None, relu=relu)
            z.backward(go)
            torch.cuda.synchronize()
        finally:
            Y._BN_LEAN[0] = True
        outs.append((z.detach(), x.grad, w.grad, beta.grad,
                     res.grad if has_res else None, gamma.grad))
    a, b = outs
    for i in range(5):
        if a[i] is None:
            continue
        if i == 3 and (H * W) % 4:
            # d(beta): the fp32 form of this geometry is the per-channel kernel,
            # whose fp64 partial sums are grouped differently
            assert float((a[i] - b[i]).abs().max()) <= 1e-6 * float(a[i].abs().max())
            continue
        assert torch.equal(a[i], b[i]), i
    sc = float(a[5].abs().max())
    assert float((a[5] - b[5]).abs().max()) <= 1e-2 * sc
    assert not torch.equal(a[5], b[5])  # the lean form did run


def test_trunk_c8_scope_outputs_only_images(bf16_mode):
    """Round 6, layers.trunk_c8_scope: a lean ConvBnActFn returns an fp32
    placeholder that is never written; the next conv takes the C8 image as its
    operand and as its residual, the backward takes the ReLU mask from it.  Against
    the same two layers outside the scope: the first layer's image is identical, the
    second differs only through the bf16 residual (<= one bf16 step of the
    residual), gradients agree to bf16 operand precision in the L2 norm; a consumer
    that would read the fp32 values raises."""
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    N, C, H, W = 2, 64, 20, 28
    lv = ((H, W), )
    base = [torch.randn(N, C, H * W, generator=g)] + [
        t for _ in range(2) for t in (torch.randn(C, C, 3, 3, generator=g) * 0.05,
                                      torch.rand(C, generator=g) + 0.5,
                                      torch.randn(C, generator=g) * 0.2)]
    mean = (torch.randn(C, generator=g) * 0.1).to(dev)
    var = (torch.rand(C, generator=g) + 0.5).to(dev)
    go = torch.randn(N, C, H * W, generator=g).to(dev)
    outs = []
    for scoped in (False, True):
        t = [b.to(dev).requires_grad_(True) for b in base]
        Y.to_c8(t[0])
        import contextlib
        with (Y.trunk_c8_scope() if scoped else contextlib.nullcontext()):
            a, _ = Y.conv_bn_act(t[0], t[1], t[2], t[3], mean, var, 1e-5, 1, 1, lv)
            b, _ = Y.conv_bn_act(a, t[4], t[5], t[6], mean, var, 1e-5, 1, 1, lv,
                                 residual=a)
        assert Y._unwritten(a) == scoped and Y._unwritten(b) == scoped
        ia, ib = Y._c8_cached(a).clone(), Y._c8_cached(b).clone()
        b.backward(go)
        torch.cuda.synchronize()
        outs.append((ia, ib, [v.grad.clone() for v in t]))
        if scoped:
            b._ld_c8 = None  # the image is gone: nothing may fall back to fp32
            with pytest.raises(L.LdError, match='only as a C8 image'):
                Y.to_c8(b)
    (ia0, ib0, g0), (ia1, ib1, g1) = outs
    assert torch.equal(ia0, ia1)
    f0, f1 = ib0.float(), ib1.float()
    assert float((f0 - f1).abs().max()) <= 2.0 ** -7 * float(f0.abs().max())
    # (a residual that moves by a bf16 step flips the ReLU mask of the ~0.1 % of
    # cells within that step of zero: those entries of dx move by a whole gradient
    # term, sqrt(1e-3) ~ 3 % of the norm with this test's white-noise gradient --
    # measured 2.2 %; the whole-step effect is test_bf16_train_step_vs_fp32_golden's)
    for u, v in zip(g0, g1):
        assert float((u - v).norm()) <= 5e-2 * float(u.norm())


def test_padded_data_gradient_for_68_output_channels(bf16_mode):
    """Round 6: the data gradient of a conv whose Cout is not a multiple of 16
    (gfl_reg, 68 corner logits) takes the bf16 kernel on a zero-padded dY (the
    weight image is zero-padded to 16 channels anyway) instead of the fp32 kernel.
    Against the fp32 kernel (LD_PAD_DGRAD=0): equal to bf16 operand rounding; the
    weight and bias gradients do not change at all."""
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    levels = ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))
    P = sum(h * w for h, w in levels)
    base = [torch.randn(2, 256, P, generator=g),
            torch.randn(68, 256, 3, 3, generator=g) * 0.02,
            torch.randn(68, generator=g)]
    go = torch.randn(2, 68, P, generator=g).to(dev)
    outs = []
    for pad in (False, True):
        Y._PAD_DGRAD[0] = pad
        try:
            x, w, b = (t.to(dev).requires_grad_(True) for t in base)
            y, _ = Y.conv2d(x, w, b, 1, 1, levels)
            y.backward(go)
            torch.cuda.synchronize()
        finally:
            Y._PAD_DGRAD[0] = True
        outs.append((y.detach(), x.grad, w.grad, b.grad))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert not torch.equal(a[1], b[1])
    assert float((a[1] - b[1]).norm()) <= 1e-2 * float(a[1].norm())


def test_bf16_wgrad_vectorised_loads_same_bits(bf16_mode, monkeypatch):
    """The 16-byte-load variants of the wave-private bf16 weight gradient (dY
    always when Pout % 4 == 0, X too for 1x1 stride-1 convs) build the same LDS
    image as the per-element kernel: identical weight gradients."""
    from ld_amd import layers as Y
    dev = _dev()
    ran = 0
    for case in BF16_CASES:
        name, N, cin, cout, k, stride, pad, levels = case
        P = sum(h * w for h, w in levels)
        g = torch.Generator().manual_seed(len(name) + cout)
        x = torch.randn(N, cin, P, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
        outs = []
        for vec in ('0', '1', 'y'):  # per-element, default, dY-only vectorised
            monkeypatch.setenv('LD_CONV_BF16_WGRAD_VEC', vec)
            wd = w.clone().requires_grad_(True)
            y, _ = Y.conv2d(x, wd, None, stride, pad, levels)
            go = torch.Generator().manual_seed(3)
            y.backward(torch.randn(y.shape, generator=go).to(dev))
            outs.append(wd.grad)
        assert torch.equal(outs[0], outs[1]), name
        assert torch.equal(outs[0], outs[2]), name
        ran += 1
    assert ran >= 8
