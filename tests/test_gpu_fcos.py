"""GPU tests (-m gpu) of the FCOS-GFL / LDFCOS head:
ld_fcos_targets bit-exact against the REFERENCE's point targets, the fused loss
block with LD_LOSS_ATSS | LD_LOSS_FCOS against the reference's LDFCOSHead.loss
tables and gradients (tests/golden/lossblock_fcos.npz)."""
import numpy as np
import pytest
import torch

from test_oracle_atss import check_grads, inputs
from test_oracle_fcos import CASES, reference_targets

pytestmark = pytest.mark.gpu


def _head(dev, ld=True):
    from ld_amd.registry import build_head
    cfg = dict(
        type='LDFCOSHead' if ld else 'FCOSGFLHead', num_classes=80,
        in_channels=256, stacked_convs=4, feat_channels=256,
        strides=[8, 16, 32, 64, 128],
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=1.0),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True,
                             loss_weight=1.0),
        norm_on_bbox=False, centerness_on_reg=True, dcn_on_last_conv=False,
        center_sampling=True, conv_bias=True)
    if ld:
        cfg.update(
            loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                         loss_weight=0.25, T=10),
            loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                         loss_weight=10, T=2))
    return build_head(cfg).to(dev)


@pytest.mark.parametrize('name', CASES)
def test_fcos_targets_bit_exact(golden, name):
    dev = torch.device('cuda:0')
    g = golden['lossblock_fcos']
    batch, sizes, hi = inputs(g, name)
    head = _head(dev)
    t = head.get_targets_batched(sizes,
                                 [b.to(dev) for b in batch['gt_bboxes']],
                                 [l.to(dev) for l in batch['gt_labels']], dev)
    N = len(batch['gt_bboxes'])
    rl, rb = reference_targets(g, name, sizes, N)
    labels = t['labels'].cpu().numpy()
    vlr = t['vlr'].cpu().numpy()
    bt = t['bbox_targets'].cpu().numpy()
    np.testing.assert_array_equal(labels, np.where(rl == 81, 80, rl))
    np.testing.assert_array_equal(vlr > 0, rl == 81)
    assigned = rl < 80
    np.testing.assert_array_equal(bt[assigned], rb[assigned])
    counts = t['counts'].cpu().numpy()
    assert counts[N + 2 * 5] == assigned.sum()
    assert (t['label_weights'] == 1).all() and not t['im'].any()


@pytest.mark.parametrize('name', CASES)
def test_ldfcos_loss_vs_reference(golden, name):
    from ld_amd.heads import ATSS_LOSS_KEYS
    dev = torch.device('cuda:0')
    g = golden['lossblock_fcos']
    batch, sizes, hi = inputs(g, name)
    head = _head(dev)
    dv = {k: [t.to(dev).requires_grad_(k in ('cls', 'reg', 'ctr'))
              for t in v] for k, v in hi.items()}
    losses = head.loss(dv['cls'], dv['reg'], dv['ctr'],
                       [b.to(dev) for b in batch['gt_bboxes']],
                       [l.to(dev) for l in batch['gt_labels']],
                       (dv['t_cls'], dv['t_reg'], None), batch['img_metas'])
    assert list(losses.keys()) == ATSS_LOSS_KEYS
    table = torch.stack([torch.stack(losses[k]) for k in ATSS_LOSS_KEYS])
    table.sum().backward()
    got = table.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, g[name + '_losses'], rtol=1e-4, atol=1e-4)
    grads = {k: [t.grad.cpu().numpy() for t in dv[k]]
             for k in ('cls', 'reg', 'ctr')}
    check_grads(g, name, grads, 5e-4, 5e-8)


def test_fcos_gfl_head_forward_and_plain_loss():
    dev = torch.device('cuda:0')
    head = _head(dev, ld=False)
    assert {'conv_cls.weight', 'conv_reg.weight', 'conv_centerness.bias',
            'cls_convs.0.conv.weight', 'reg_convs.3.gn.weight',
            'scales.4.scale'} <= set(head.state_dict())
    from ld_amd import synthetic
    pad = (128, 160)
    sizes = synthetic.level_shapes(pad)
    feats = [torch.randn(2, 256, h, w, device=dev) for h, w in sizes]
    cls, reg, ctr = head(feats)
    assert [tuple(c.shape) for c in ctr] == [(2, 1, h, w) for h, w in sizes]
    batch = synthetic.synthetic_batch(2, pad, pad, [3, 0], 5)  # one empty image
    losses = head.forward_train(
        feats, batch['img_metas'], [b.to(dev) for b in batch['gt_bboxes']],
        [l.to(dev) for l in batch['gt_labels']])
    assert list(losses.keys()) == ['loss_cls', 'loss_bbox', 'loss_centerness']
    tot = sum(sum(v) for v in losses.values())
    tot.backward()
    assert torch.isfinite(tot)


@pytest.mark.parametrize('name', ['tiny', 'small'] if __import__('os').environ.get('LD_TEST_FULL') == '1' else ['small'])
def test_ld_fcos_train_step_vs_reference(golden, name):
    """Whole detector step of configs/ld/ld_r50_fcos_r101_1x.py (caffe-style
    ResNets with frozen BN, FPN with relu_before_extra_convs, LDFCOSHead <-
    FCOS-GFL R101) against the reference's loss table, gradient norms and
    gradient projections."""
    from ld_amd import model_zoo, synthetic
    from ld_amd.heads import ATSS_LOSS_KEYS
    from ld_amd.registry import build_detector
    dev = torch.device('cuda:0')
    g = golden['e2e_fcos']
    cfg = g[name + '_cfg']
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
    num_gt = [int(x) for x in g[name + '_num_gt']]
    det = build_detector(model_zoo.ld_fcos_detector(50, 101))
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
    table = torch.stack([torch.stack(losses[k]) for k in ATSS_LOSS_KEYS])
    loss, _ = det._parse_losses(losses)
    loss.backward()
    got = table.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, g[name + '_losses'], rtol=1e-4, atol=1e-4)
    names = [str(k) for k in g[name + '_grad_names']]
    params = dict(det.named_parameters())
    bad, off = [], []
    for k, rn, pr in zip(names, g[name + '_grad_norms'],
                         g[name + '_grad_proj']):
        assert params[k].grad is not None, k
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
    got_names = sorted(k for k, p in params.items() if p.grad is not None)
    assert got_names == sorted(names)
