"""GPU (-m gpu): the other imitation regions of SURVEY.md section 8f-4 through
the HIP path -- 'fitnet' (centre-inside mask from the targets kernel) and
'gibox' (ld_gi_region: GI scores/boxes + top-10 greedy NMS per level) for
LDHead and LDv2Head -- against golden vectors from the reference
(tests/golden/imitation.npz).  Selected cells exact, losses within 1e-4."""
import numpy as np
import pytest
import torch

from ld_amd import synthetic

pytestmark = pytest.mark.gpu

LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im']

CASES = [('fitnet_small', 'v1', 'fitnet'), ('fitnet_c2', 'v1', 'fitnet'),
         ('gibox_small', 'v1', 'gibox'), ('gibox_c2', 'v1', 'gibox'),
         ('gibox_v2_small', 'v2', 'gibox')]


def _head(kind, method, dev):
    from ld_amd import model_zoo
    from ld_amd.config import ConfigDict
    from ld_amd.registry import build_head
    if kind == 'v1':
        cfg = dict(model_zoo.ld_detector(50, 101, imitation_method=method,
                                         loss_im_weight=2.0)['bbox_head'])
    else:
        cfg = dict(model_zoo.ldv2_detector(50, 101, imitation_method=method,
                                           loss_im_weight=2.0)['bbox_head'])
    cfg.update(train_cfg=ConfigDict.wrap(model_zoo._TRAIN_CFG),
               test_cfg=ConfigDict.wrap(model_zoo._TEST_CFG))
    head = build_head(cfg)
    if kind == 'v2':
        head.load_state_dict(synthetic.seeded_state_dict(head.state_dict(),
                                                         seed=5))
    return head.to(dev)


@pytest.mark.parametrize('name,kind,method', CASES, ids=[c[0] for c in CASES])
def test_imitation_modes_vs_reference(golden, name, kind, method):
    from ld_amd import layers as Y
    dev = torch.device('cuda:0')
    g = golden['imitation']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed, hseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4], cfg[5]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                         num_classes=80 if kind == 'v1' else 81)
    head = _head(kind, method, dev)
    cls = [t.to(dev).requires_grad_(True) for t in hi['cls']]
    reg = [t.to(dev).requires_grad_(True) for t in hi['reg']]
    xs = [t.to(dev).requires_grad_(True) for t in hi['x']]
    gtb = [b.to(dev) for b in batch['gt_bboxes']]
    gtl = [l.to(dev) for l in batch['gt_labels']]
    tx = [t.to(dev) for t in hi['t_x']]
    if kind == 'v1':
        losses = head.loss(cls, reg, gtb, gtl,
                           ([t.to(dev) for t in hi['t_cls']],
                            [t.to(dev) for t in hi['t_reg']]), xs, tx,
                           batch['img_metas'])
    else:
        c0, c2 = head.reg_conf[0], head.reg_conf[2]
        scores = []
        for c, r in zip(cls, reg):
            n, ch, h, w = c.shape
            s3, _ = Y.QualityFn.apply(r.reshape(n, 68, h * w),
                                      c.reshape(n, ch, h * w), c0.weight,
                                      c0.bias, c2.weight, c2.bias)
            scores.append(s3.view(n, ch, h, w))
        losses = head.loss(scores, reg, cls, gtb, gtl,
                           (None, [t.to(dev) for t in hi['t_reg']],
                            [t.to(dev) for t in hi['t_cls']]), xs, tx,
                           batch['img_metas'])
    table = torch.stack([torch.stack(losses[k]) for k in LOSS_KEYS])
    sum(sum(v) for v in losses.values()).backward()
    torch.cuda.synchronize()
    got = table.detach().cpu().numpy()
    print(name, 'max abs loss err', np.abs(got - g[name + '_losses']).max())
    np.testing.assert_allclose(got, g[name + '_losses'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(
        [float(t.grad.double().abs().sum()) for t in xs],
        g[name + '_gx_abs_sum'], rtol=2e-4, atol=1e-7)
    np.testing.assert_array_equal(
        [int((t.grad.abs().sum(1) > 0).sum()) for t in xs],
        g[name + '_gx_nonzero_rows'])
    if method == 'gibox':
        # the selected cells themselves: mask (N, A) level-major vs the
        # reference's row indices n * H*W + r
        im = head.last_targets['im'].cpu().numpy()
        off = 0
        for l, (h, w) in enumerate(sizes):
            want = np.sort(g[f'{name}_gi_idx_{l}'])
            n_idx, r_idx = np.nonzero(im[:, off:off + h * w])
            np.testing.assert_array_equal(np.sort(n_idx * h * w + r_idx), want)
            off += h * w


def test_decouple_raises_like_the_reference():
    dev = torch.device('cuda:0')
    head = _head('v1', 'decouple', dev)
    sizes = synthetic.level_shapes((160, 224))
    hi = synthetic.synthetic_head_inputs(1, sizes, seed=1)
    b = synthetic.synthetic_batch(1, (160, 224), (160, 224), [2], 3)
    mv = lambda ts: [t.to(dev) for t in ts]  # noqa: E731
    with pytest.raises(NotImplementedError, match='decouple'):
        head.loss(mv(hi['cls']), mv(hi['reg']), mv(b['gt_bboxes']),
                  mv(b['gt_labels']), (mv(hi['t_cls']), mv(hi['t_reg'])),
                  mv(hi['x']), mv(hi['t_x']), b['img_metas'])
