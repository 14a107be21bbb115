"""LDATSSHead: the numpy restatement of
LDATSSHead.loss (oracle/ld_oracle.py ld_atss_loss_block; ld_atss.py:44-250 over
atss_gfl_head.py) against the loss tables and gradients the REFERENCE produced
(tests/golden/lossblock_atss.npz, oracle/gen_golden.py gen_lossblock_atss)."""
import os
import sys

import numpy as np
import pytest

from ld_amd import synthetic

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'oracle'))
import ld_oracle as O  # noqa: E402

CASES = ['small', 'small_crowd', 'c2', 'c2_crowd']


def inputs(g, name):
    cfg = g[name + '_cfg']
    pad, img_shape = tuple(cfg[:2]), tuple(cfg[2:4])
    bseed, hseed = int(cfg[4]), int(cfg[5])
    num_gt = [int(x) for x in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt,
                                      bseed)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed)
    hi['ctr'] = synthetic.synthetic_centerness(len(num_gt), sizes, seed=hseed)
    return batch, sizes, hi


def check_grads(g, name, grads, rtol, atol):
    for k in ('cls', 'reg', 'ctr'):
        for l, gr in enumerate(grads[k]):
            gr = np.asarray(gr)
            a = np.abs(gr.astype(np.float64)).sum()
            np.testing.assert_allclose(a, g[f'{name}_g{k}_abs_sum'][l],
                                       rtol=1e-4, atol=1e-7)
            if f'{name}_g{k}_{l}' in g:
                np.testing.assert_allclose(gr, g[f'{name}_g{k}_{l}'],
                                           rtol=rtol, atol=atol)
            else:
                flat = gr.reshape(-1)
                idx = np.arange(0, flat.size, 1009)
                np.testing.assert_allclose(flat[idx],
                                           g[f'{name}_g{k}_{l}_sample'],
                                           rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', CASES)
def test_atss_lossblock_vs_reference(golden, name):
    g = golden['lossblock_atss']
    batch, sizes, hi = inputs(g, name)
    hi = {k: [t.numpy() for t in v] for k, v in hi.items()}
    t = O.get_targets(sizes, batch['img_metas'],
                      [b.numpy() for b in batch['gt_bboxes']],
                      [l.numpy() for l in batch['gt_labels']])
    out = O.ld_atss_loss_block(hi['cls'], hi['reg'], hi['ctr'], hi['t_cls'],
                               hi['t_reg'], t)
    np.testing.assert_allclose(out['losses'], g[name + '_losses'], rtol=2e-5,
                               atol=2e-6)
    check_grads(g, name, out['grads'], 2e-4, 2e-8)


def test_focal_and_centerness_hand_cases():
    x = np.array([0.0, 2.0, -3.0], np.float32)
    f1, d1 = O.focal_elements(x, np.ones(3), 0.25)
    f0, d0 = O.focal_elements(x, np.zeros(3), 0.25)
    p = 1 / (1 + np.exp(-x.astype(np.float64)))
    np.testing.assert_allclose(f1, 0.25 * (1 - p) ** 2 * -np.log(p), rtol=1e-6)
    np.testing.assert_allclose(f0, 0.75 * p ** 2 * -np.log(1 - p), rtol=1e-6)
    eps = 1e-3
    for t, d in ((np.ones(3), d1), (np.zeros(3), d0)):
        num = (O.focal_elements(x + eps, t)[0].astype(np.float64) -
               O.focal_elements(x - eps, t)[0]) / (2 * eps)
        np.testing.assert_allclose(d, num, rtol=2e-3, atol=1e-5)
    anchors = np.array([[-32, -32, 32, 32]], np.float32)   # centre (0, 0)
    gts = np.array([[-10, -20, 30, 20]], np.float32)
    np.testing.assert_allclose(O.centerness_target(anchors, gts),
                               [np.sqrt((10 / 30) * (20 / 20))], rtol=1e-6)
