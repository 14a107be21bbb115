"""GPU tests (-m gpu) of the registry-level components used on their own with
the reference's call signatures and tensor layouts -- the tests the reference
would have had for this path (its tests/test_models/test_losses.py and
tests/test_assigner.py cover neither GFL nor ATSS) -- checked against golden
vectors produced by the reference code (tests/golden/kat_losses.npz,
targets.npz) and the numpy oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _t(a, dev, grad=False):
    t = torch.as_tensor(np.asarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def test_kd_kl_loss_module(golden):
    from ld_amd import build_loss
    dev = _dev()
    g = golden['kat_losses']
    pred = _t(g['kl_pred'], dev, True)
    soft, w = _t(g['kl_soft'], dev), _t(g['kl_w'], dev)
    kl = build_loss(dict(type='KnowledgeDistillationKLDivLoss',
                         loss_weight=0.25, T=10))
    none = kl(pred, soft, reduction_override='none')
    np.testing.assert_allclose(none.detach().cpu().numpy(), g['kat1_none'],
                               rtol=2e-5, atol=1e-8)
    loss = kl(pred, soft, weight=w, avg_factor=4.0)
    np.testing.assert_allclose(float(loss), g['kat1_mean'], rtol=2e-5)
    loss.backward()
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g['kat1_grad'],
                               rtol=2e-4, atol=1e-9)
    # KD settings (T=2, lw 10) and the plain-mean / sum reductions
    pred2 = _t(g['kl_pred'], dev, True)
    kd = build_loss(dict(type='KnowledgeDistillationKLDivLoss',
                         loss_weight=10, T=2))
    l2 = kd(pred2, soft, weight=torch.ones(4, device=dev), avg_factor=4)
    np.testing.assert_allclose(float(l2), g['kat1b_mean'], rtol=2e-5)
    l2.backward()
    np.testing.assert_allclose(pred2.grad.cpu().numpy(), g['kat1b_grad'],
                               rtol=2e-4, atol=1e-7)
    s = kl(pred.detach(), soft, reduction_override='sum')
    np.testing.assert_allclose(float(s), g['kat1_none'].sum(), rtol=2e-5)
    with pytest.raises(ValueError):  # losses/utils.py:52-54
        kl(pred.detach(), soft, avg_factor=2.0, reduction_override='sum')
    # alias registered for the side heads' default type string
    alias = build_loss(dict(type='LocalizationDistillationLoss',
                            loss_weight=0.25, T=10))
    np.testing.assert_allclose(
        float(alias(pred.detach(), soft, weight=w, avg_factor=4.0)),
        g['kat1_mean'], rtol=2e-5)


def test_dfl_qfl_giou_modules(golden):
    from ld_amd import build_loss
    dev = _dev()
    g = golden['kat_losses']
    w = _t(g['kl_w'], dev)
    # DFL (KAT2)
    pred = _t(g['kl_pred'], dev, True)
    dfl = build_loss(dict(type='DistributionFocalLoss', loss_weight=0.25))
    lab = _t(g['dfl_label'], dev)
    np.testing.assert_allclose(
        dfl(pred, lab, reduction_override='none').detach().cpu().numpy(),
        g['kat2_none'], rtol=1e-5)
    loss = dfl(pred, lab, weight=w, avg_factor=4.0)
    np.testing.assert_allclose(float(loss), g['kat2_mean'], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g['kat2_grad'],
                               rtol=2e-4, atol=1e-7)
    # QFL (KAT4)
    cp = _t(g['qfl_pred'], dev, True)
    labels, score = _t(g['qfl_labels'], dev), _t(g['qfl_score'], dev)
    qfl = build_loss(dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0,
                          loss_weight=1.0))
    np.testing.assert_allclose(
        qfl(cp, (labels, score),
            reduction_override='none').detach().cpu().numpy(),
        g['kat4_none'], rtol=1e-5)
    loss = qfl(cp, (labels, score), weight=torch.ones(6, device=dev),
               avg_factor=2.5)
    np.testing.assert_allclose(float(loss), g['kat4_mean'], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(cp.grad.cpu().numpy(), g['kat4_grad'],
                               rtol=2e-4, atol=1e-7)
    # GIoU (KAT5) incl. the all-zero-weight early-out (iou_loss.py:341-342)
    b1 = _t(g['giou_b1'], dev, True)
    b2, gw = _t(g['giou_b2'], dev), _t(g['giou_w'], dev)
    giou = build_loss(dict(type='GIoULoss', loss_weight=2.0))
    np.testing.assert_allclose(
        giou(b1, b2, reduction_override='none').detach().cpu().numpy(),
        g['kat5_none'], rtol=1e-5)
    loss = giou(b1, b2, weight=gw, avg_factor=1.0)
    np.testing.assert_allclose(float(loss), g['kat5_mean'], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(b1.grad.cpu().numpy(), g['kat5_grad'],
                               rtol=2e-4, atol=1e-7)
    # (as in the reference the early-out multiplies pred by weight, so the
    # weight must broadcast against (n, 4))
    assert float(giou(b1.detach(), b2,
                      weight=torch.zeros((3, 4), device=dev))) == 0
    # IMLoss
    im = build_loss(dict(type='IMLoss', loss_weight=2.0))
    a = _t(g['im_a'], dev, True)
    l = im(a, _t(g['im_b'], dev))
    np.testing.assert_allclose(float(l), g['im_loss'], rtol=1e-5)
    l.backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g['im_grad'], rtol=1e-4,
                               atol=1e-8)
    # empty inputs (no positives on a level)
    z = dfl(torch.zeros((0, 17), device=dev), torch.zeros(0, device=dev))
    assert z.numel() <= 1


def test_integral_overlaps_transforms(golden):
    from ld_amd.core import (BboxOverlaps2D, bbox2distance, bbox_overlaps,
                             distance2bbox)
    from ld_amd.heads import Integral
    dev = _dev()
    g = golden['kat_losses']
    pred = _t(g['kl_pred'], dev, True)
    integ = Integral(16).to(dev)
    e = integ(pred.reshape(1, 68))
    np.testing.assert_allclose(e.detach().cpu().numpy(), g['kat3_integral'],
                               rtol=1e-6)
    (e * torch.tensor([1., 2., 3., 4.], device=dev)).sum().backward()
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g['kat3_grad'],
                               rtol=2e-4, atol=1e-6)
    b1, b2 = _t(g['giou_b1'], dev), _t(g['giou_b2'], dev)
    np.testing.assert_allclose(
        bbox_overlaps(b1, b2, is_aligned=True).cpu().numpy(),
        g['kat6_iou_aligned'], rtol=1e-6)
    calc = BboxOverlaps2D()
    for mode in ('iou', 'iof', 'giou', 'diou'):
        np.testing.assert_array_equal(calc(b1, b2, mode).cpu().numpy(),
                                      g['kat6_pair_' + mode])
    assert calc(b1[:0], b2).shape == (0, 3)
    pts, dist = _t(g['d2b_points'], dev), _t(g['d2b_dist'], dev)
    bx = distance2bbox(pts, dist)
    np.testing.assert_array_equal(bx.cpu().numpy(), g['d2b_out'])
    np.testing.assert_array_equal(
        bbox2distance(pts, bx, max_dis=16).cpu().numpy(), g['b2d_out'])


def test_atss_assigner_reference_signature(golden):
    """ATSSAssigner.assign / get_vlr_region with explicit anchors, the way
    LDHead._get_target_single calls them (ld_head.py:505-517)."""
    from ld_amd.registry import (build_anchor_generator, build_assigner,
                                 build_sampler)
    dev = _dev()
    g = golden['targets']
    ag = build_anchor_generator(
        dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8,
             scales_per_octave=1, strides=[8, 16, 32, 64, 128]))
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    anchors = torch.cat(ag.grid_anchors(sizes, device=dev))
    flags = ag.valid_flags(sizes, (64, 64), device=dev)
    assert [int(f.sum()) for f in flags] == [64, 16, 4, 1, 1]
    nl = [64, 16, 4, 1, 1]
    assigner = build_assigner(dict(type='ATSSAssigner', topk=9))
    gts, gl = _t(g['kat7b_gt'], dev), _t(g['kat7b_labels'], dev)
    ar = assigner.assign(anchors, nl, gts, None, gl)
    np.testing.assert_array_equal(ar.gt_inds.cpu().numpy(),
                                  g['kat7b_gt_inds'])
    np.testing.assert_array_equal(ar.max_overlaps.cpu().numpy(),
                                  g['kat7b_max_overlaps'])
    pos = ar.gt_inds > 0
    np.testing.assert_array_equal(
        ar.labels[pos].cpu().numpy(),
        g['kat7b_labels'][g['kat7b_gt_inds'][g['kat7b_gt_inds'] > 0] - 1])
    assert int((ar.labels[~pos] != -1).sum()) == 0
    vlr = assigner.get_vlr_region(anchors, nl, gts, None, gl)
    np.testing.assert_array_equal(vlr.cpu().numpy(), g['kat7b_vlr'])
    # PseudoSampler: sorted unique index lists (pseudo_sampler.py:24-41)
    sr = build_sampler(dict(type='PseudoSampler')).sample(ar, anchors, gts)
    np.testing.assert_array_equal(sr.pos_inds.cpu().numpy(),
                                  np.nonzero(g['kat7b_gt_inds'])[0])
    np.testing.assert_array_equal(
        sr.pos_gt_bboxes.cpu().numpy(),
        g['kat7b_gt'][g['kat7b_gt_inds'][g['kat7b_gt_inds'] > 0] - 1])
    # no GT: everything background, zeros for the VLR map
    empty = assigner.assign(anchors, nl, gts[:0], None, gl[:0])
    assert int(empty.gt_inds.abs().sum()) == 0
    assert float(assigner.get_vlr_region(anchors, nl, gts[:0]).sum()) == 0


def test_gfl_head_plain_loss_vs_oracle(golden):
    """GFLHead.loss (no distillation) = QFL + GIoU + DFL of the fused block."""
    import ld_oracle as O
    from ld_amd import build_head, synthetic
    from ld_amd.config import ConfigDict
    dev = _dev()
    head = build_head(dict(
        type='GFLHead', num_classes=80, in_channels=256,
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        train_cfg=ConfigDict(assigner=dict(type='ATSSAssigner', topk=9),
                             allowed_border=-1, pos_weight=-1, debug=False),
        test_cfg=None)).to(dev)
    pad = (160, 224)
    batch = synthetic.synthetic_batch(2, pad, pad, [3, 1], 11)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(2, sizes, seed=101)
    cls = [t.to(dev).requires_grad_(True) for t in hi['cls']]
    reg = [t.to(dev).requires_grad_(True) for t in hi['reg']]
    losses = head.loss(cls, reg, [b.to(dev) for b in batch['gt_bboxes']],
                       [l.to(dev) for l in batch['gt_labels']],
                       batch['img_metas'])
    assert sorted(losses) == ['loss_bbox', 'loss_cls', 'loss_dfl']
    hin = {k: [x.numpy() for x in v] for k, v in hi.items()}
    t = O.get_targets(sizes, batch['img_metas'],
                      [b.numpy() for b in batch['gt_bboxes']],
                      [l.numpy() for l in batch['gt_labels']])
    o = O.ld_loss_block(hin['cls'], hin['reg'], hin['cls'], hin['reg'],
                        hin['x'], hin['x'], t,
                        dict(lw_ld=0, lw_ld_vlr=0, lw_kd=0, lw_im=0))
    for i, k in enumerate(('loss_cls', 'loss_bbox', 'loss_dfl')):
        got = np.array([float(v) for v in losses[k]])
        np.testing.assert_allclose(got, o['losses'][i], rtol=1e-4, atol=1e-5)
    sum(sum(v) for v in losses.values()).backward()
    for l in range(5):
        np.testing.assert_allclose(cls[l].grad.cpu().numpy(),
                                   o['grads']['cls'][l], rtol=5e-4, atol=1e-7)
        np.testing.assert_allclose(reg[l].grad.cpu().numpy(),
                                   o['grads']['reg'][l], rtol=5e-4, atol=1e-7)
