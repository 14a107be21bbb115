"""GPU parity (-m gpu) of the ResNeXt-101 32x4d teacher (BASELINE config 5):
the grouped-conv kernel (ld_amd/csrc/gconv.hip) and the whole backbone against
the reference's own ResNeXt run on CPU (tests/golden/resnext.npz from
mmdet/models/backbones/resnext.py:11-153 via oracle/gen_golden.py), plus the
whole LDv2 R50 <- X101 step against tests/golden/e2e_v2_r3.npz.
Tolerance: element-wise 2e-4 of the tensor scale (fp32 summation order);
losses 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ld_amd import synthetic

pytestmark = pytest.mark.gpu

LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im']


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


GCONV_CASES = [  # N, C, groups, H, W, k, stride
    (2, 128, 32, 40, 56, 3, 1), (1, 256, 32, 21, 27, 3, 2),
    (2, 512, 32, 13, 17, 3, 1), (1, 1024, 32, 7, 9, 3, 1),
    (1, 128, 32, 1, 1, 3, 1), (2, 256, 32, 16, 300, 3, 1),
    (1, 512 * 9, 32, 9, 11, 1, 1),  # the grouped GEMM behind a grouped DCN
]


@pytest.mark.parametrize('case', GCONV_CASES, ids=[str(c) for c in GCONV_CASES])
@pytest.mark.parametrize('epilogue', [False, True])
def test_grouped_conv_vs_torch_cpu(case, epilogue):
    """ld_gconv_forward against torch's CPU grouped conv2d (the op the
    reference's ResNeXt runs), with and without the folded BN + ReLU."""
    from ld_amd import layers as Y
    N, C, G, H, W, k, s = case
    dev = _dev()
    g = torch.Generator().manual_seed(C * 7 + H)
    cout = C if k == 3 else C // 9
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(cout, C // G, k, k, generator=g) / (C // G * k * k)**0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, s, k // 2, 1, G)
    if epilogue:
        ref = F.relu(ref * scale.double()[None, :, None, None] +
                     shift.double()[None, :, None, None])
    y3, lv = Y.gconv_forward(
        x.to(dev).reshape(N, C, H * W), w.to(dev), G, s, k // 2, ((H, W), ),
        scale.to(dev) if epilogue else None,
        shift.to(dev) if epilogue else None, epilogue)
    assert lv == ((ref.shape[2], ref.shape[3]), )
    got = y3.view(ref.shape).cpu().double()
    sc = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max())
    assert err <= 2e-5 * sc + 1e-7, (err, sc)


@pytest.mark.parametrize('case', ['x101_small', 'x101_mid', 'x50_odd'])
def test_resnext_features_vs_reference(golden, case):
    """All four stage outputs element-wise against the reference's ResNeXt."""
    from ld_amd import model_zoo
    from ld_amd.registry import build_backbone
    g = golden['resnext']
    depth, n, h, w, seed, step = [int(v) for v in g[case + '_cfg']]
    dev = _dev()
    net = build_backbone(model_zoo._x101_backbone(depth))
    net.load_state_dict(synthetic.seeded_state_dict(net.state_dict(), seed=seed))
    net.to(dev).eval()
    gen = torch.Generator().manual_seed(seed + 100)
    x = torch.randn(n, 3, h, w, generator=gen).to(dev)
    with torch.no_grad():
        outs = net(x)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(int(v) for v in g[f'{case}_shape{i}'])
        got = o.cpu().numpy().reshape(-1)[::step].astype(np.float64)
        ref = g[f'{case}_out{i}'].astype(np.float64)
        sc = float(np.abs(ref).max()) + 1e-12
        err = float(np.abs(got - ref).max())
        print(case, 'stage', i, 'err', err, 'scale', sc)
        assert err <= 2e-4 * sc, (case, i, err, sc)


def test_grouped_dcn_zero_offsets_equal_grouped_conv():
    """A grouped DeformConv2dPack whose offset conv is zero (its init) IS the
    grouped conv on the same weight: ties the grouped-DCN path (im2col + grouped
    GEMM) to the pinned grouped-conv kernel.  (DCN itself: parity unpinned.)"""
    from ld_amd.cnn import DeformConv2dPack, GroupedConv2d
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for stride in (1, 2):
        d = DeformConv2dPack(512, 512, 3, stride=stride, padding=1, groups=32)
        c = GroupedConv2d(512, 512, 3, stride=stride, padding=1, groups=32)
        w = torch.randn(512, 16, 3, 3, generator=g) / 12.0
        with torch.no_grad():
            d.weight.copy_(w)
            c.weight.copy_(w)
        d.to(dev)
        c.to(dev)
        x = torch.randn(2, 512, 13, 18, generator=g).to(dev)
        with torch.no_grad():
            a, b = d(x), c(x)
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * sc


@pytest.mark.parametrize('name', ['v2x_small_r50', 'v2x_c2_r50', 'v2_c2_r50'])
def test_ldv2_step_vs_reference_r3(golden, name):
    """Whole LDv2 steps against the reference (oracle/gen_golden.py
    gen_e2e_v2_r3): the R101-teacher step at the BASELINE config-2 size, and
    BASELINE config 5's R50 <- X101 composition at two sizes.  Loss table /
    log_vars 1e-4, every parameter gradient by norm (5e-3) and by two random
    projections (sign / order / layout sensitive)."""
    from ld_amd import build_detector, model_zoo
    dev = _dev()
    g = golden['e2e_v2_r3']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    x101 = name.startswith('v2x')
    det = build_detector(model_zoo.ldv2_x101_detector() if x101
                         else model_zoo.ldv2_detector(50, 101))
    det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(), seed=1))
    tsd = det.teacher_model.state_dict()
    assert list(tsd.keys()) == [str(k) for k in g[name + '_teacher_keys']]
    det.teacher_model.load_state_dict(synthetic.seeded_state_dict(tsd, seed=2))
    det.to(dev).train()
    d = dict(img=batch['img'].to(dev), img_metas=batch['img_metas'],
             gt_bboxes=[b.to(dev) for b in batch['gt_bboxes']],
             gt_labels=[l.to(dev) for l in batch['gt_labels']])
    with torch.no_grad():
        tx = det.teacher_model.extract_feat(d['img'])
    np.testing.assert_allclose(
        [float(f.double().abs().mean()) for f in tx],
        g[name + '_teacher_feat_abs_mean'], rtol=2e-4)
    losses = det(**d)
    table = torch.stack([torch.stack(losses[k]) for k in LOSS_KEYS])
    loss, log_vars = det._parse_losses(losses)
    loss.backward()
    torch.cuda.synchronize()
    got = table.detach().cpu().numpy().astype(np.float64)
    ref = g[name + '_losses']
    print(name, 'max abs loss err', np.abs(got - ref).max())
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    for k, r in zip(LOSS_KEYS + ['loss'], g[name + '_log_vars']):
        np.testing.assert_allclose(log_vars[k], r, rtol=1e-4, atol=1e-4,
                                   err_msg=k)
    names = [str(k) for k in g[name + '_grad_names']]
    params = dict(det.named_parameters())
    bad = []
    for k, r, pr in zip(names, g[name + '_grad_norms'], g[name + '_grad_proj']):
        assert params[k].grad is not None, k
        gflat = params[k].grad.double().reshape(-1).cpu().numpy()
        got_n = float(np.linalg.norm(gflat))
        if not np.isclose(got_n, r, rtol=5e-3, atol=1e-6):
            bad.append((k, 'norm', got_n, r))
        for sd, want in zip((0, 1), pr):
            gp = float(gflat @ synthetic.grad_probe(gflat.size, sd))
            if abs(gp - want) > 1e-2 * r + 1e-6:
                bad.append((k, f'proj{sd}', gp, want))
    assert not bad, f'{len(bad)} gradient checks off, first: {bad[:5]}'
    if name == 'v2_c2_r50':
        # ... and element-wise against the reference's own gradients (round 4)
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _gradcheck import check_grad_samples
        worst = check_grad_samples(golden, name, params)
        print(name, 'vs reference, max|err|/max|g| top 5:', [(f'{r:.1e}', k) for r, k in worst[:5]])
        off = [(k, float(params[k].grad.double().norm()), float(r))
               for k, r in zip(names, g[name + '_grad_norms'])
               if not np.isclose(float(params[k].grad.double().norm()), r, rtol=1e-3,
                                 atol=1e-7)]
        assert not off, f'{len(off)} grad norms off at 1e-3, first: {off[:5]}'
