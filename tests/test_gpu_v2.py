"""GPU parity tests (-m gpu) of SURVEY.md row R-V2: GFocalHead's fused
distribution-guided quality branch (quality.hip) and LDv2Head's loss block
variant, against
 (1) golden vectors produced by the reference's own GFocalHead / LDv2Head
     (tests/golden/lossblock_v2.npz, e2e_v2.npz; oracle/gen_golden.py), and
 (2) the CPU oracle (oracle/net_oracle.py quality_tail, ldv2_*), itself pinned
     on those goldens by tests/test_oracle_v2.py.
Tolerance: fp32 losses within 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from ld_amd import synthetic

pytestmark = pytest.mark.gpu

LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im']


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _close(got, ref, rtol, atol_rel, what):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = float(np.abs(ref).max()) + 1e-30
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * scale,
                               err_msg=what)


def _head_params(seed=5):
    shapes = {'reg_conf.0.weight': (64, 20, 1, 1), 'reg_conf.0.bias': (64, ),
              'reg_conf.2.weight': (1, 64, 1, 1), 'reg_conf.2.bias': (1, )}
    return synthetic.seeded_state_dict(
        {k: torch.zeros(v) for k, v in shapes.items()}, seed=seed)


@pytest.mark.parametrize('N,C,P', [(2, 81, 1000), (1, 81, 257), (3, 11, 64)])
def test_quality_branch_fwd_bwd_vs_oracle(N, C, P):
    """quality.hip forward + backward (incl. the reg_conf parameter gradients)
    against torch autograd through the oracle's restatement of
    gfocal_head.py:201-217."""
    import net_oracle as NO
    from ld_amd import layers as Y
    dev = _dev()
    g = torch.Generator().manual_seed(N * 1000 + P)
    reg = torch.randn(N, 68, P, generator=g) * 3
    cf = torch.randn(N, C, P, generator=g) * 1.5 - 2
    sd = {k: v.clone().requires_grad_(True) for k, v in _head_params().items()}
    gout = torch.randn(N, C, P, generator=g)
    # oracle (as an (N, C, P, 1) map)
    r_ = reg.clone().requires_grad_(True)
    c_ = cf.clone().requires_grad_(True)
    score, q = NO.quality_tail(sd, c_[..., None], r_[..., None], prefix='')
    score.backward(gout[..., None])
    rd = reg.to(dev).requires_grad_(True)
    cd = cf.to(dev).requires_grad_(True)
    pd = [sd[k].detach().to(dev).requires_grad_(True) for k in sd]
    sc, qd = Y.QualityFn.apply(rd, cd, *pd)
    _close(qd.cpu().numpy(), q[:, 0, :, 0].detach().numpy(), 2e-5, 1e-6, 'quality')
    _close(sc.detach().cpu().numpy(), score[..., 0].detach().numpy(), 2e-5,
           1e-6, 'cls_score')
    sc.backward(gout.to(dev))
    _close(cd.grad.cpu().numpy(), c_.grad.numpy(), 1e-4, 1e-6, 'g cls_feat')
    _close(rd.grad.cpu().numpy(), r_.grad.numpy(), 1e-4, 2e-6, 'g reg')
    for p, k in zip(pd, sd):
        _close(p.grad.cpu().numpy(), sd[k].grad.numpy(), 2e-4, 2e-6, 'g ' + k)


def _v2_head(dev, lw_im=2.0):
    from ld_amd import model_zoo
    from ld_amd.registry import build_head
    from ld_amd.config import ConfigDict
    cfg = dict(model_zoo.ldv2_detector(50, 101, loss_im_weight=lw_im)
               ['bbox_head'])
    cfg.update(train_cfg=ConfigDict.wrap(model_zoo._TRAIN_CFG),
               test_cfg=ConfigDict.wrap(model_zoo._TEST_CFG))
    head = build_head(cfg)
    head.load_state_dict(synthetic.seeded_state_dict(head.state_dict(),
                                                     seed=5))
    return head.to(dev)


@pytest.mark.parametrize('name', ['v2_small', 'v2_small_crowd', 'v2_c2'])
def test_ldv2_lossblock_parity(golden, name):
    """LDv2Head.loss (quality kernel -> fused loss block -> quality backward)
    on seeded tower outputs against the reference's LDv2Head.loss."""
    from ld_amd import layers as Y
    dev = _dev()
    g = golden['lossblock_v2']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed, hseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4], cfg[5]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                         num_classes=81)
    head = _v2_head(dev)
    cf = [t.to(dev).requires_grad_(True) for t in hi['cls']]
    rg = [t.to(dev).requires_grad_(True) for t in hi['reg']]
    xs = [t.to(dev).requires_grad_(True) for t in hi['x']]
    c0, c2 = head.reg_conf[0], head.reg_conf[2]
    scores = []
    for c, r in zip(cf, rg):
        n, ch, h, w = c.shape
        s3, _ = Y.QualityFn.apply(r.reshape(n, 68, h * w),
                                  c.reshape(n, ch, h * w), c0.weight, c0.bias,
                                  c2.weight, c2.bias)
        scores.append(s3.view(n, ch, h, w))
    teacher = (None, [t.to(dev) for t in hi['t_reg']],
               [t.to(dev) for t in hi['t_cls']])
    losses = head.loss(scores, rg, cf,
                       [b.to(dev) for b in batch['gt_bboxes']],
                       [l.to(dev) for l in batch['gt_labels']], teacher, xs,
                       [t.to(dev) for t in hi['t_x']], batch['img_metas'])
    assert list(losses.keys()) == LOSS_KEYS
    table = torch.stack([torch.stack(losses[k]) for k in LOSS_KEYS])
    sum(sum(v) for v in losses.values()).backward()
    torch.cuda.synchronize()
    got = table.detach().cpu().numpy()
    print(name, 'max abs loss err', np.abs(got - g[name + '_losses']).max())
    np.testing.assert_allclose(got, g[name + '_losses'], rtol=1e-4, atol=1e-4)
    for k in ('reg_conf.0.weight', 'reg_conf.0.bias', 'reg_conf.2.weight',
              'reg_conf.2.bias'):
        p = dict(head.named_parameters())[k]
        _close(p.grad.cpu().numpy(), g[f'{name}_gparam_{k}'], 5e-4, 5e-6,
               'grad ' + k)
    for key, ts in (('cls', cf), ('reg', rg), ('x', xs)):
        _close([float(t.grad.double().abs().sum()) for t in ts],
               g[f'{name}_g{key}_abs_sum'], 2e-4, 1e-7, f'|g{key}|')
        for l, t in enumerate(ts):
            full = f'{name}_g{key}_{l}'
            gt_ = t.grad.cpu().numpy()
            if full in g.files:
                _close(gt_, g[full], 2e-4, 2e-6, full)
            else:
                flat = gt_.reshape(-1)
                _close(flat[np.arange(0, flat.size, 1009)],
                       g[full + '_sample'], 2e-4, 2e-6, full)


@pytest.mark.parametrize('name', ['v2_tiny_r50', 'v2_small_r50'] if __import__('os').environ.get('LD_TEST_FULL') == '1' else ['v2_small_r50'])
def test_ldv2_train_step_vs_reference_golden(golden, name):
    """Whole LDv2 step (R50 LDv2Head student <- R101 GFocalHead teacher)
    against the reference run from configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py
    (imitation 'finegrained')."""
    from ld_amd import build_detector, model_zoo
    dev = _dev()
    g = golden['e2e_v2']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    det = build_detector(model_zoo.ldv2_detector(50, 101))
    det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(), seed=1))
    det.teacher_model.load_state_dict(
        synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2))
    det.to(dev).train()
    d = dict(img=batch['img'].to(dev), img_metas=batch['img_metas'],
             gt_bboxes=[b.to(dev) for b in batch['gt_bboxes']],
             gt_labels=[l.to(dev) for l in batch['gt_labels']])
    losses = det(**d)
    assert list(losses.keys()) == LOSS_KEYS
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
    for k, r in zip(names, g[name + '_grad_norms']):
        assert params[k].grad is not None, k
        got_n = float(params[k].grad.double().norm())
        if not np.isclose(got_n, r, rtol=5e-3, atol=1e-6):
            bad.append((k, got_n, r))
    assert not bad, f'{len(bad)} grad norms off, first: {bad[:5]}'
    # one optimizer step through the train engine (direct gradient sinks of the
    # quality branch included) must run
    from ld_amd.train import SGDTrainer
    tr = SGDTrainer(det, lr=0.0025)
    out = tr.step(d)
    torch.cuda.synchronize()
    assert np.isfinite(float(out['loss']))


# ---------------------------------------------------------------- config 4 ---
DCN_CASES = [  # name, N, Cin, Cout, H, W, stride, offset scale
    ('c3_like', 2, 128, 128, 20, 28, 1, 1.5),
    ('c3_first_s2', 1, 128, 128, 21, 27, 2, 2.0),
    ('c5_like', 2, 512, 512, 7, 11, 1, 4.0),
    ('small_odd', 1, 64, 96, 5, 6, 1, 6.0),
]


@pytest.mark.parametrize('case', DCN_CASES, ids=[c[0] for c in DCN_CASES])
def test_dcn_forward_vs_oracle(case):
    """DeformConv2dPack (offset conv -> ld_deform_im2col -> 1x1 MFMA GEMM with
    the folded BN/ReLU epilogue) against the torch-CPU restatement of DCNv1
    (oracle/dcn_oracle.py).  Offsets are large enough to leave the map.
    Tolerance 2e-4 rel + 2e-5 of the tensor scale (fp32 summation order)."""
    import dcn_oracle as D
    from ld_amd import layers as Y
    from ld_amd.cnn import DeformConv2dPack
    name, N, cin, cout, H, W, stride, oscale = case
    dev = _dev()
    g = torch.Generator().manual_seed(len(name) * 13 + cin)
    x = torch.randn(N, cin, H, W, generator=g)
    m = DeformConv2dPack(cin, cout, 3, stride=stride, padding=1)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) / (9 * cin)**0.5)
        m.conv_offset.weight.copy_(
            torch.randn(m.conv_offset.weight.shape, generator=g) * oscale /
            (9 * cin)**0.5)
        m.conv_offset.bias.copy_(torch.randn(18, generator=g) * 0.7)
    ref, off_ref = D.dcn_pack_forward(x, m.weight.detach(),
                                      m.conv_offset.weight.detach(),
                                      m.conv_offset.bias.detach(), stride, 1)
    assert float(off_ref.abs().max()) > 2.0  # samples do leave the 3x3 window
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    res = torch.randn(ref.shape, generator=g)
    m = m.to(dev)
    with torch.no_grad():
        y = m(x.to(dev))
        _close(y.cpu().numpy(), ref.numpy(), 2e-4, 2e-5, name + ' plain')
        y3, lv = m.forward3_fused(
            x.to(dev).reshape(N, cin, H * W), ((H, W), ), scale.to(dev),
            shift.to(dev), res.to(dev).reshape(N, cout, -1), True)
    fused = torch.relu(ref * scale[None, :, None, None] +
                       shift[None, :, None, None] + res)
    _close(y3.cpu().numpy().reshape(fused.shape), fused.numpy(), 2e-4, 2e-5,
           name + ' fused epilogue')
    with pytest.raises(NotImplementedError):
        m(x.to(dev).requires_grad_(True))


def test_config4_teacher_step():
    """R101 student <- R101-DCN(c3-c5) teacher (BASELINE config 4 at a small
    size): the LD step runs, and with the DCN offsets at their initial value 0
    the teacher equals the plain R101 teacher with the same weights."""
    from ld_amd import build_detector, model_zoo
    dev = _dev()
    cfg = model_zoo.ld_detector(50, 101, loss_im_weight=2.0)
    cfg['teacher_config']['model']['backbone'].update(
        dcn=dict(type='DCN', deform_groups=1, fallback_on_stride=False),
        stage_with_dcn=(False, True, True, True))
    det = build_detector(cfg)
    plain = build_detector(model_zoo.ld_detector(50, 101, loss_im_weight=2.0))
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(plain.teacher_model.state_dict(), seed=2)
    det.load_state_dict(ssd)
    plain.load_state_dict(ssd)
    plain.teacher_model.load_state_dict(tsd)
    missing = det.teacher_model.load_state_dict(tsd, strict=False)
    assert all('conv_offset' in k for k in missing.missing_keys)
    assert len(missing.missing_keys) == 2 * (4 + 23 + 3)
    assert not missing.unexpected_keys
    for k, v in det.teacher_model.state_dict().items():
        if 'conv_offset' in k:
            v.zero_()
    det.to(dev).train()
    plain.to(dev).train()
    b = synthetic.synthetic_batch(2, (128, 150), (128, 160), [3, 2], 21)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    t1 = torch.stack([torch.stack(v) for v in det(**d).values()])
    t0 = torch.stack([torch.stack(v) for v in plain(**d).values()])
    np.testing.assert_allclose(t1.detach().cpu().numpy(),
                               t0.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    # nonzero offsets change the teacher (and only the distillation terms)
    for k, v in det.teacher_model.state_dict().items():
        if k.endswith('conv_offset.bias'):
            v.fill_(0.6)
    t2 = torch.stack([torch.stack(v) for v in det(**d).values()])
    diff = (t2 - t1).abs().sum(1).detach().cpu().numpy()
    assert diff[LOSS_KEYS.index('loss_ld')] > 1e-4
    assert diff[LOSS_KEYS.index('loss_cls')] == 0.0
    loss, _ = det._parse_losses(det(**d))
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
