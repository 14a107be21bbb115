"""CPU: the oracle's other imitation regions --
'fitnet' (anchor centre inside a GT) and 'gibox' (GI boxes: max class-score gap
+ greedy NMS, top 10 per level) -- against golden vectors the reference's
LDHead / LDv2Head produced (oracle/gen_golden.py gen_imitation; get_gi_region
executed on CPU with `.cuda()` neutralised and torchvision's nms restated)."""
import numpy as np
import pytest

from ld_amd import synthetic


def _inputs(golden, name, channels):
    g = golden['imitation']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed, hseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4], cfg[5]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                         num_classes=channels)
    return g, batch, sizes, hi


@pytest.mark.parametrize('name', ['fitnet_small', 'fitnet_c2'])
def test_fitnet_region_and_loss(golden, name):
    import ld_oracle as O
    g, batch, sizes, hi = _inputs(golden, name, 80)
    t = O.get_targets(sizes, batch['img_metas'],
                      [b.numpy() for b in batch['gt_bboxes']],
                      [l.numpy() for l in batch['gt_labels']],
                      im_mode='center')
    npy = lambda ts: [x.numpy() for x in ts]  # noqa: E731
    out = O.ld_loss_block(npy(hi['cls']), npy(hi['reg']), npy(hi['t_cls']),
                          npy(hi['t_reg']), npy(hi['x']), npy(hi['t_x']), t,
                          dict(lw_im=2.0))
    np.testing.assert_allclose(out['losses'], g[name + '_losses'], rtol=2e-5,
                               atol=2e-6)
    np.testing.assert_allclose(
        [np.abs(a).sum(dtype=np.float64) for a in out['grads']['x']],
        g[name + '_gx_abs_sum'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('name,prob', [('gibox_small', False),
                                       ('gibox_c2', False),
                                       ('gibox_v2_small', True)])
def test_gibox_region_and_loss(golden, name, prob):
    import ld_oracle as O
    g, batch, sizes, hi = _inputs(golden, name, 81 if prob else 80)
    npy = lambda ts: [x.detach().numpy() for x in ts]  # noqa: E731
    t = O.get_targets(sizes, batch['img_metas'],
                      [b.numpy() for b in batch['gt_bboxes']],
                      [l.numpy() for l in batch['gt_labels']],
                      im_mode='center')
    if prob:
        import net_oracle as NO
        import torch
        from test_oracle_v2 import _head_sd
        sd = _head_sd()
        scores = [NO.quality_tail(sd, c, r, prefix='')[0]
                  for c, r in zip(hi['cls'], hi['reg'])]
        cls = npy(scores)
    else:
        cls = npy(hi['cls'])
    idxs, mask = O.gi_region(cls, npy(hi['reg']), npy(hi['t_cls']),
                             npy(hi['t_reg']), prob=prob)
    for l, idx in enumerate(idxs):
        np.testing.assert_array_equal(idx, g[f'{name}_gi_idx_{l}'])
    t = dict(t)
    t['im'] = mask
    out = O.ld_loss_block(cls, npy(hi['reg']), None if prob else npy(hi['t_cls']),
                          npy(hi['t_reg']), npy(hi['x']), npy(hi['t_x']), t,
                          dict(lw_im=2.0),
                          kd=(npy(hi['cls']), npy(hi['t_cls'])) if prob else None)
    np.testing.assert_allclose(out['losses'], g[name + '_losses'], rtol=2e-5,
                               atol=2e-6)
    np.testing.assert_array_equal(
        [int((np.abs(a).sum(1) > 0).sum()) for a in out['grads']['x']],
        g[name + '_gx_nonzero_rows'])
