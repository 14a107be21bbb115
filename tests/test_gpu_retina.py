"""GPU tests (-m gpu) of the RetinaGFL / LDRetina head:
9 anchors per cell.  The MaxIoU + VLR targets kernel (ld_retina_targets)
bit-exact against the REFERENCE's LDRetinaHead.get_targets, the fused loss block
with LD_LOSS_RETINA (FocalLoss with the ignore band, GIoU on decoded boxes, LD
over the 68 corner logits, 0.03 x VLR-LD, KD) against the reference's
LDRetinaHead.loss (tests/golden/lossblock_retina.npz): loss table within 1e-4,
gradients of the summed table element-wise; then the registry-level head and a
whole detector step of configs/ld/ld_retina_r50_1x.py."""
import os

import numpy as np
import pytest
import torch

from test_oracle_retina import CASES, check_grads, inputs, oracle_targets

pytestmark = pytest.mark.gpu


def _head(dev, ld=True):
    from ld_amd.config import ConfigDict
    from ld_amd.registry import build_head
    cfg = dict(
        type='LDRetinaHead' if ld else 'RetinaGFLHead', num_classes=80,
        in_channels=256, stacked_convs=4, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4,
                              scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                              strides=[8, 16, 32, 64, 128]),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                        target_means=[.0, .0, .0, .0],
                        target_stds=[1.0, 1.0, 1.0, 1.0]),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        reg_decoded_bbox=True,
        train_cfg=ConfigDict.wrap(dict(
            assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5,
                          neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1),
            allowed_border=-1, pos_weight=-1, debug=False)))
    if ld:
        cfg.update(
            loss_ld=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=5,
                         T=10),
            loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                         loss_weight=10, T=8))
    return build_head(cfg).to(dev)


def _to_reference_order(t, sizes, B):
    """pseudo-image layout (N * B, A[, 4]) -> per level (N, A_l * B[, 4]) in the
    reference's (cell, base anchor) order."""
    NB = t.shape[0]
    N = NB // B
    t = t.reshape((N, B) + tuple(t.shape[1:]))
    out, off = [], 0
    for h, w in sizes:
        lv = t[:, :, off:off + h * w]
        off += h * w
        lv = lv.permute(0, 2, 1, *range(3, lv.dim()))
        out.append(lv.reshape((N, h * w * B) + tuple(lv.shape[3:])))
    return out


@pytest.mark.parametrize('name', CASES)
def test_retina_targets_bit_exact_vs_reference(golden, name):
    dev = torch.device('cuda:0')
    g = golden['lossblock_retina']
    batch, sizes, _ = inputs(g, name)
    head = _head(dev)
    t = head.get_targets_batched(
        sizes, batch['img_metas'], [b.to(dev) for b in batch['gt_bboxes']],
        [l.to(dev) for l in batch['gt_labels']], dev, want_gt_inds=True)
    B = 9
    assert int(t['counts'][-1]) == int(g[name + '_num_total_pos'])
    lab = _to_reference_order(t['labels'], sizes, B)
    lw = _to_reference_order(t['label_weights'], sizes, B)
    vlr = _to_reference_order(t['vlr'], sizes, B)
    bt = _to_reference_order(t['bbox_targets'], sizes, B)
    gi = _to_reference_order(t['gt_inds'], sizes, B)
    o = oracle_targets(batch, sizes)
    s = 0
    for l in range(len(sizes)):
        assert np.array_equal(lab[l].cpu().numpy(), g[f'{name}_labels_{l}'])
        assert np.array_equal(lw[l].cpu().numpy(),
                              g[f'{name}_label_weights_{l}'])
        # the VLR MASK is exact; the value is an IoU (same arithmetic)
        v = vlr[l].cpu().numpy()
        assert np.array_equal(v > 0, g[f'{name}_vlr_{l}'] > 0)
        np.testing.assert_allclose(v, g[f'{name}_vlr_{l}'], rtol=1e-6,
                                   atol=1e-7)
        pos = lab[l] < 80
        assert np.array_equal(bt[l][pos].cpu().numpy(),
                              g[f'{name}_bbox_pos_{l}'])
        n = o['num_level'][l]
        assert np.array_equal(gi[l].cpu().numpy(), o['gt_inds'][:, s:s + n])
        s += n
    # level counts of positives
    L_ = len(sizes)
    NB = t['labels'].shape[0]
    want = [int((g[f'{name}_labels_{l}'] < 80).sum()) for l in range(L_)]
    assert t['counts'][NB:NB + L_].tolist() == want


@pytest.mark.parametrize('name', CASES)
def test_ldretina_loss_vs_reference(golden, name):
    from ld_amd.heads import RETINA_LOSS_KEYS
    dev = torch.device('cuda:0')
    g = golden['lossblock_retina']
    batch, sizes, hi = inputs(g, name)
    head = _head(dev)
    dv = {k: [t.to(dev).requires_grad_(k in ('cls', 'reg')) for t in v]
          for k, v in hi.items()}
    losses = head.loss(dv['cls'], dv['reg'],
                       [b.to(dev) for b in batch['gt_bboxes']],
                       [l.to(dev) for l in batch['gt_labels']],
                       (dv['t_cls'], dv['t_reg']), batch['img_metas'])
    assert list(losses.keys()) == RETINA_LOSS_KEYS
    table = torch.stack([torch.stack(losses[k]) for k in RETINA_LOSS_KEYS])
    table.sum().backward()
    got = table.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, g[name + '_losses'], rtol=1e-4, atol=1e-4)
    grads = {k: [t.grad.cpu().numpy() for t in dv[k]] for k in ('cls', 'reg')}
    check_grads(g, name, grads, 5e-4, 5e-8)


def test_ldretina_nonunit_upstream_vs_oracle():
    """Weighted sum of the loss entries (the rerun-with-upstream path) against
    the oracle's analytic gradients with the same per-key weights."""
    import ld_oracle as O
    from ld_amd import synthetic
    dev = torch.device('cuda:0')
    pad, num_gt = (160, 224), [4, 2]
    batch = synthetic.synthetic_batch(2, pad, pad, num_gt, 31)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(2, sizes, seed=131, num_anchors=9)
    head = _head(dev)
    dv = {k: [t.to(dev).requires_grad_(k in ('cls', 'reg')) for t in v]
          for k, v in hi.items()}
    losses = head.loss(dv['cls'], dv['reg'],
                       [b.to(dev) for b in batch['gt_bboxes']],
                       [l.to(dev) for l in batch['gt_labels']],
                       (dv['t_cls'], dv['t_reg']), batch['img_metas'])
    (2.0 * sum(losses['loss_ld']) + 0.5 * sum(losses['loss_ld_vlr']) +
     3.0 * sum(losses['loss_bbox'])).backward()
    t = oracle_targets(batch, sizes)
    hn = {k: [t_.numpy() for t_ in v] for k, v in hi.items()}
    ref = O.ld_retina_loss_block(
        hn['cls'], hn['reg'], hn['t_cls'], hn['t_reg'], t,
        hp=dict(lw_cls=0, lw_kd=0, lw_ld=10.0, vlr_factor=0.03 * 0.25,
                lw_bbox=6.0))
    for l in range(5):
        np.testing.assert_allclose(dv['reg'][l].grad.cpu().numpy(),
                                   ref['grads']['reg'][l], rtol=5e-4,
                                   atol=2e-8)
        assert float(dv['cls'][l].grad.abs().max()) == 0.0


def test_retina_gfl_head_forward_and_plain_loss():
    """RetinaGFLHead: conv + ReLU towers (no norm), 9 x 80 / 9 x 68 channels
    from the reference's parameter names; its own loss (no teacher) = the cls /
    bbox rows; gradients reach the towers through the ReLU."""
    from ld_amd import synthetic
    dev = torch.device('cuda:0')
    head = _head(dev, ld=False)
    keys = set(head.state_dict())
    assert {'atss_cls.weight', 'atss_cls.bias', 'atss_reg.weight',
            'cls_convs.0.conv.weight', 'cls_convs.0.conv.bias',
            'reg_convs.3.conv.bias', 'integral.project'} <= keys
    assert not any('gn' in k or 'scales' in k for k in keys)
    head.init_weights()
    pad = (128, 160)
    sizes = synthetic.level_shapes(pad)
    feats = [torch.randn(2, 256, h, w, device=dev) for h, w in sizes]
    cls, reg = head(feats)
    assert [tuple(c.shape) for c in cls] == [(2, 720, h, w) for h, w in sizes]
    assert [tuple(c.shape) for c in reg] == [(2, 612, h, w) for h, w in sizes]
    # towers vs plain torch
    import torch.nn.functional as F
    with torch.no_grad():
        x = feats[1]
        for m in head.cls_convs:
            x = F.relu(F.conv2d(x, m.conv.weight, m.conv.bias, padding=1))
        want = F.conv2d(x, head.atss_cls.weight, head.atss_cls.bias, padding=1)
        fused = head(feats)[0][1]   # no-grad path: conv+bias+ReLU in one launch
    for got in (cls[1], fused):
        assert (got - want).abs().max() <= 2e-4 * want.abs().max() + 1e-6
    batch = synthetic.synthetic_batch(2, pad, pad, [3, 2], 5)
    losses = head.forward_train(
        feats, batch['img_metas'], [b.to(dev) for b in batch['gt_bboxes']],
        [l.to(dev) for l in batch['gt_labels']])
    assert list(losses.keys()) == ['loss_cls', 'loss_bbox']
    tot = sum(sum(v) for v in losses.values())
    tot.backward()
    assert torch.isfinite(tot)
    for m in (head.cls_convs[0].conv, head.reg_convs[0].conv, head.atss_reg):
        assert m.weight.grad is not None and m.bias.grad is not None
        assert float(m.weight.grad.abs().sum()) > 0


def test_max_iou_assigner_api_vs_oracle():
    """MaxIoUAssigner.assign(bboxes, gt_bboxes, gt_labels=...) on an arbitrary
    box list: gt_inds / labels equal the oracle's (reference semantics: -1
    ignore, 0 negative, g + 1 positive)."""
    import ld_oracle as O
    from ld_amd.registry import build_assigner
    dev = torch.device('cuda:0')
    a = build_assigner(dict(type='MaxIoUAssigner', pos_iou_thr=0.5,
                            neg_iou_thr=0.4, min_pos_iou=0,
                            ignore_iof_thr=-1))
    g = torch.Generator().manual_seed(3)
    ctr = torch.rand(700, 2, generator=g) * 200
    wh = torch.rand(700, 2, generator=g) * 60 + 4
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    gts = torch.tensor([[20., 30, 90, 100], [100, 100, 180, 150],
                        [5, 5, 15, 12], [150, 20, 190, 70]])
    labels = torch.tensor([3, 7, 7, 0])
    res = a.assign(boxes.to(dev), gts.to(dev), gt_labels=labels.to(dev))
    want = O.max_iou_assign(boxes.numpy(), gts.numpy())
    assert np.array_equal(res.gt_inds.cpu().numpy(), want)
    assert set(np.unique(want)) >= {-1, 0, 1}
    wl = np.where(want > 0, labels.numpy()[np.maximum(want, 1) - 1], -1)
    assert np.array_equal(res.labels.cpu().numpy(), wl)
    empty = a.assign(boxes.to(dev), gts[:0].to(dev))
    assert int(empty.gt_inds.abs().sum()) == 0


@pytest.mark.parametrize(
    'name', ['tiny', 'small'] if os.environ.get('LD_TEST_FULL') == '1'
    else ['small'])
def test_ld_retina_train_step_vs_reference(golden, name):
    """Whole detector step of configs/ld/ld_retina_r50_1x.py (LDRetinaHead R50
    student <- RetinaGFL R101 teacher, FPN extra convs on the input,
    output_feature=False) against the reference's loss table, gradient norms
    and gradient projections."""
    from ld_amd import model_zoo, synthetic
    from ld_amd.heads import RETINA_LOSS_KEYS
    from ld_amd.registry import build_detector
    dev = torch.device('cuda:0')
    g = golden['e2e_retina']
    cfg = g[name + '_cfg']
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
    num_gt = [int(x) for x in g[name + '_num_gt']]
    det = build_detector(model_zoo.ld_retina_detector(50, 101))
    assert list(det.state_dict().keys()) == \
        [str(k) for k in g[name + '_student_keys']]
    assert list(det.teacher_model.state_dict().keys()) == \
        [str(k) for k in g[name + '_teacher_keys']]
    det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(), seed=1))
    det.teacher_model.load_state_dict(synthetic.seeded_state_dict(
        det.teacher_model.state_dict(), seed=2))
    det.to(dev)
    det.train()
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt,
                                      bseed)
    losses = det(img=batch['img'].to(dev), img_metas=batch['img_metas'],
                 gt_bboxes=[b.to(dev) for b in batch['gt_bboxes']],
                 gt_labels=[l.to(dev) for l in batch['gt_labels']])
    assert list(losses.keys()) == RETINA_LOSS_KEYS
    table = torch.stack([torch.stack(losses[k]) for k in RETINA_LOSS_KEYS])
    loss, log_vars = det._parse_losses(losses)
    loss.backward()
    got = table.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, g[name + '_losses'], rtol=1e-4, atol=1e-4)
    names = [str(k) for k in g[name + '_grad_names']]
    params = dict(det.named_parameters())
    bad, off = [], []
    for k, rn, pr in zip(names, g[name + '_grad_norms'],
                         g[name + '_grad_proj']):
        gflat = params[k].grad.double().reshape(-1).cpu().numpy()
        if not np.isclose(np.linalg.norm(gflat), rn, rtol=5e-3, atol=1e-6):
            bad.append((k, float(np.linalg.norm(gflat)), float(rn)))
        for sd in (0, 1):
            probe = synthetic.grad_probe(gflat.size, sd)
            tol = 4 * 5e-3 * rn * np.linalg.norm(probe) / \
                np.sqrt(gflat.size) + 1e-7
            if abs(float(gflat @ probe) - pr[sd]) > tol:
                off.append((k, sd))
    assert not bad, f'{len(bad)} grad norms off: {bad[:4]}'
    assert not off, f'{len(off)} grad projections off: {off[:4]}'
    for k, p in params.items():
        if not p.requires_grad:
            assert p.grad is None, k


def test_ld_retina_sgd_steps():
    """Two SGDTrainer steps (fused unit-upstream backward, arena, SGD) on the
    LD-Retina detector, fp32 and bf16 (C8-only teacher trunk feeding the FPN's
    on-input extra conv): every trainable parameter receives a gradient."""
    from ld_amd import layers as Y
    from ld_amd import model_zoo, synthetic
    from ld_amd.registry import build_detector
    from ld_amd.train import SGDTrainer
    dev = torch.device('cuda:0')
    for mode in ('fp32', 'bf16'):
        det = build_detector(model_zoo.ld_retina_detector(50, 50))
        det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(),
                                                        seed=1))
        det.teacher_model.load_state_dict(synthetic.seeded_state_dict(
            det.teacher_model.state_dict(), seed=2))
        det.to(dev)
        det.train()
        os.environ['LD_CHECK_GRADS'] = '1'
        Y.set_precision(mode)
        try:
            tr = SGDTrainer(det, lr=1e-5)
            b = synthetic.synthetic_batch(2, (128, 160), (128, 160), [3, 2], 9)
            d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                     gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                     gt_labels=[x.to(dev) for x in b['gt_labels']])
            l0 = float(tr.step(d)['loss'])
            l1 = float(tr.step(d)['loss'])
        finally:
            os.environ.pop('LD_CHECK_GRADS', None)
            Y.set_precision('fp32')
        assert np.isfinite(l0) and np.isfinite(l1), mode
