"""GPU tests (-m gpu) of round 5's fused gradient accumulation: the data gradient
with an addend summed in the GEMM epilogue (ld_conv_dgrad_acc and its bf16
forms), the FPN upsample backward with an addend, level pack / unpack, and the
fan protocol inside whole train steps (same losses, gradients equal to the
unfused step up to the order of a handful of fp32 additions)."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # name, N, Cin, Cout, k, stride, pad, levels
    ('1x1_64_256', 2, 64, 256, 1, 1, 0, ((20, 34), )),
    ('1x1_s2_256_512', 2, 256, 512, 1, 2, 0, ((20, 34), )),
    ('1x1_s2_odd', 1, 64, 128, 1, 2, 0, ((13, 21), )),
    ('3x3_64_64', 2, 64, 64, 3, 1, 1, ((24, 40), )),
    ('3x3_s2_128', 2, 128, 128, 3, 2, 1, ((26, 42), )),
    ('3x3_256_256_levels', 2, 256, 256, 3, 1, 1,
     ((20, 28), (10, 14), (5, 7), (3, 4), (2, 2))),
    ('3x3_256_68_levels', 1, 256, 68, 3, 1, 1,
     ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))),
    ('1x1_1024_256', 2, 1024, 256, 1, 1, 0, ((13, 21), )),
]


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'bf16_c8'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_dgrad_acc(case, mode):
    """dx = dgrad(dy) + addend from ONE launch == the plain launch followed by an
    fp32 add, bit for bit (the epilogue adds the addend to the finished
    accumulator: one rounding, as the separate add has); also in place."""
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    name, N, cin, cout, k, stride, pad, levels = case
    bf16 = mode != 'fp32'
    if bf16 and cout % 16 != 0:
        pytest.skip('bf16 data gradient reduces over Cout: multiple of 16')
    if mode == 'bf16_c8' and cout % 32 != 0:
        pytest.skip('C8 operand image: Cout multiple of 32')
    g = torch.Generator().manual_seed(len(name) * 31 + cout)
    d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
    dy = torch.randn(N, cout, d.Pout, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cout * k * k)**0.5).to(dev)
    addend = torch.randn(N, cin, d.Pin, generator=g).to(dev)
    st = L.stream_ptr(dev)
    prev = Y.get_precision()
    Y.set_precision('bf16' if bf16 else 'fp32')
    try:
        _, wt = Y.weight_images(w, True, bf16=bf16, need_fwd=False)
        if mode == 'bf16_c8':
            dyin = Y.to_c8(dy)
            plain, acc = lib.ld_conv_bf16_dgrad_c8, lib.ld_conv_bf16_dgrad_c8_acc
        elif bf16:
            dyin = dy
            plain, acc = lib.ld_conv_bf16_dgrad, lib.ld_conv_bf16_dgrad_acc
        else:
            dyin = dy
            plain, acc = lib.ld_conv_dgrad, lib.ld_conv_dgrad_acc
        dx0 = torch.full((N, cin, d.Pin), float('nan'), device=dev)
        L.check(plain(C.byref(d), L.ptr(dyin), L.ptr(wt), L.ptr(dx0), st), 'dgrad')
        dx1 = torch.full((N, cin, d.Pin), float('nan'), device=dev)
        L.check(acc(C.byref(d), L.ptr(dyin), L.ptr(wt), L.ptr(addend), L.ptr(dx1), st),
                'dgrad_acc')
        dx2 = addend.clone()  # in place: addend == dx
        L.check(acc(C.byref(d), L.ptr(dyin), L.ptr(wt), L.ptr(dx2), L.ptr(dx2), st),
                'dgrad_acc in place')
        torch.cuda.synchronize()
    finally:
        Y.set_precision(prev)
    want = dx0 + addend
    assert torch.isfinite(dx1).all()
    assert torch.equal(dx1, want), (name, mode, float((dx1 - want).abs().max()))
    assert torch.equal(dx2, want), (name, mode, 'in place')


def test_upsample_add_backward_acc():
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    g = torch.Generator().manual_seed(3)
    for (hf, wf, hc, wc) in ((100, 168, 50, 84), (25, 42, 13, 21), (13, 21, 7, 11)):
        rows = 2 * 16
        dout = torch.randn(rows, hf * wf, generator=g).to(dev)
        addend = torch.randn(rows, hc * wc, generator=g).to(dev)
        st = L.stream_ptr(dev)
        a = torch.empty(rows, hc * wc, device=dev)
        b = torch.empty(rows, hc * wc, device=dev)
        L.check(lib.ld_upsample_add_backward(L.ptr(dout), rows, hf, wf, hc, wc,
                                             L.ptr(a), st), 'plain')
        L.check(lib.ld_upsample_add_backward_acc(L.ptr(dout), rows, hf, wf, hc, wc,
                                                 L.ptr(addend), L.ptr(b), st), 'acc')
        torch.cuda.synchronize()
        assert torch.equal(b, a + addend)


def test_pack_unpack_levels():
    from ld_amd import layers as Y
    dev = _dev()
    levels = ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))
    g = torch.Generator().manual_seed(9)
    feats = [torch.randn(2, 24, h, w, generator=g).to(dev).requires_grad_(True)
             for h, w in levels]
    with torch.no_grad():
        x3, lv = Y.pack_levels(feats)
    assert lv == levels
    assert torch.equal(x3, torch.cat([f.flatten(2) for f in feats], 2))
    # differentiable form: forward the same, backward = the level slices
    x3g, _ = Y.pack_levels([f * 1.0 for f in feats])
    go = torch.randn(x3g.shape, generator=g).to(dev)
    x3g.backward(go)
    off = 0
    for f, (h, w) in zip(feats, levels):
        assert torch.equal(f.grad, go[:, :, off:off + h * w].reshape(f.shape))
        off += h * w


def _step_grads(monkeypatch, fuse, precision, size=(256, 320)):
    from ld_amd import layers as Y
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import SGDTrainer
    dev = _dev()
    Y._FAN_ON[0] = fuse
    prev = Y.get_precision()
    Y.set_precision(precision)
    try:
        det = model_zoo.build_seeded_ld_detector(50, 101, dev)
        tr = SGDTrainer(det, lr=0.0)  # lr 0: parameters stay put, gradients are kept
        b = synthetic.synthetic_batch(2, size, size, [5, 3], 77)
        d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                 gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                 gt_labels=[x.to(dev) for x in b['gt_labels']])
        before = dict(Y.FAN_STATS)
        out = tr.step(d)
        torch.cuda.synchronize()
        stats = {k: Y.FAN_STATS[k] - before[k] for k in before}
        return (float(out['log_vars']['loss']), tr.arena.flat_grad.clone(), stats)
    finally:
        Y.set_precision(prev)
        Y._FAN_ON[0] = os.environ.get('LD_FAN_FUSE', '1') == '1'


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_train_step_fan_fused_equals_autograd_sums(monkeypatch, precision):
    """The whole LD step with the fan-out gradients summed in the consumers'
    kernels against the same step with autograd's elementwise sums: identical
    loss; every gradient equal up to the ORDER of <= 4 fp32 additions per element
    ((a + b) + c vs a + (b + c)), which the data-gradient GEMMs downstream turn
    into ~1e-6-relative differences."""
    loss0, g0, s0 = _step_grads(monkeypatch, False, precision)
    loss1, g1, s1 = _step_grads(monkeypatch, True, precision)
    assert s0['fused'] == 0
    # 13 residual blocks + 2 stage outputs x 2 + the packed head input x 2 + 4 in
    # the neck: at least 20 sums moved into kernels, none through the fallback
    assert s1['fused'] >= 20, s1
    assert s1['fallback_adds'] == 0, s1
    assert loss0 == loss1
    scale = float(g0.abs().max())
    err = float((g0 - g1).abs().max())
    # bf16: a 1e-7 difference in a gradient occasionally crosses a bf16 rounding
    # boundary of the next GEMM's operand (one 2^-9 step for that element)
    tol, ntol = (2e-5, 5e-6) if precision == 'fp32' else (5e-3, 2e-3)
    assert err <= tol * scale, (err, scale)
    assert float((g0 - g1).norm() / g0.norm()) <= ntol
