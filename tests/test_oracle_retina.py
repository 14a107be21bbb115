"""LDRetinaHead: the numpy restatement of the
9-anchor MaxIoU / VLR targets and of LDRetinaHead.loss (oracle/ld_oracle.py
retina_targets, ld_retina_loss_block; ld_retina.py over retina_gfl_head.py and
max_iou_assigner.py) against what the REFERENCE produced
(tests/golden/lossblock_retina.npz, oracle/gen_golden.py gen_lossblock_retina)."""
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
    hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                         num_anchors=9)
    return batch, sizes, hi


def oracle_targets(batch, sizes):
    return O.retina_targets(sizes, batch['img_metas'],
                            [b.numpy() for b in batch['gt_bboxes']],
                            [l.numpy() for l in batch['gt_labels']])


def check_grads(g, name, grads, rtol, atol):
    for k in ('cls', 'reg'):
        for l, gr in enumerate(grads[k]):
            gr = np.asarray(gr)
            a = np.abs(gr.astype(np.float64)).sum()
            np.testing.assert_allclose(a, g[f'{name}_g{k}_abs_sum'][l],
                                       rtol=1e-4, atol=1e-7)
            flat = gr.reshape(-1)
            want = g[f'{name}_g{k}_{l}_sample']
            step = 11 if want.size == len(range(0, flat.size, 11)) else 1009
            np.testing.assert_allclose(flat[np.arange(0, flat.size, step)],
                                       want, rtol=rtol, atol=atol)


def test_retina_anchors_bit_exact(golden):
    g = golden['lossblock_retina']
    sizes = synthetic.level_shapes((160, 224))
    for l, a in enumerate(O.retina_grid_anchors(sizes)):
        assert np.array_equal(a, g[f'anchors_{l}'])


@pytest.mark.parametrize('name', CASES)
def test_retina_targets_vs_reference(golden, name):
    """labels / label weights (ignore band = 0) bit-exact, VLR values and the
    positives' box targets equal, num_total_pos equal."""
    g = golden['lossblock_retina']
    batch, sizes, _ = inputs(g, name)
    t = oracle_targets(batch, sizes)
    assert t['num_total_pos'] == int(g[name + '_num_total_pos'])
    s = 0
    for l, n in enumerate(t['num_level']):
        sl = slice(s, s + n)
        s += n
        assert np.array_equal(t['labels'][:, sl], g[f'{name}_labels_{l}'])
        assert np.array_equal(t['label_weights'][:, sl],
                              g[f'{name}_label_weights_{l}'])
        np.testing.assert_array_equal(t['vlr'][:, sl], g[f'{name}_vlr_{l}'])
        assert np.array_equal(t['bbox_targets'][:, sl][t['pos_mask'][:, sl]],
                              g[f'{name}_bbox_pos_{l}'])


@pytest.mark.parametrize('name', CASES)
def test_retina_lossblock_vs_reference(golden, name):
    g = golden['lossblock_retina']
    batch, sizes, hi = inputs(g, name)
    hi = {k: [t.numpy() for t in v] for k, v in hi.items()}
    t = oracle_targets(batch, sizes)
    out = O.ld_retina_loss_block(hi['cls'], hi['reg'], hi['t_cls'],
                                 hi['t_reg'], t)
    np.testing.assert_allclose(out['losses'], g[name + '_losses'], rtol=2e-5,
                               atol=2e-6)
    check_grads(g, name, out['grads'], 2e-4, 2e-8)


def test_max_iou_assign_hand_case():
    """pos >= 0.5, neg < 0.4, the band between ignored; every gt keeps its
    best anchor(s) even below the thresholds, a later gt overriding."""
    anchors = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [20, 20, 30, 30],
                        [0, 0, 10, 4.4], [50, 50, 60, 60]], np.float32)
    gts = np.array([[0, 0, 10, 10], [21, 21, 40, 40]], np.float32)
    gi = O.max_iou_assign(anchors, gts)
    # anchor 1: IoU 0.5 -> positive; anchor 3: 0.44 -> ignored; anchor 2 is
    # gt 1's best (IoU ~0.21 < 0.4) -> low-quality positive; anchor 4: negative
    assert gi.tolist() == [1, 1, 2, -1, 0]
    assert O.max_iou_assign(anchors, np.zeros((0, 4))).tolist() == [0] * 5


def test_reference_retina_anchor_kat():
    """The reference's own known-answer test for the ratios x scales anchor
    generator (tests/test_anchor.py:190-288, test_retina_anchor): base anchors
    (allclose, as there), valid-pixel counts [57600, 14400, 3600, 900, 225] at
    (640, 640), nine base anchors per level -- for BOTH the product class
    (ld_amd.core.AnchorGenerator, host code) and the oracle."""
    import torch
    from ld_amd.registry import build_anchor_generator
    ag = build_anchor_generator(dict(
        type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3,
        ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128]))
    level0 = np.array([[-22.6274, -11.3137, 22.6274, 11.3137],
                       [-28.5088, -14.2544, 28.5088, 14.2544],
                       [-35.9188, -17.9594, 35.9188, 17.9594],
                       [-16.0000, -16.0000, 16.0000, 16.0000],
                       [-20.1587, -20.1587, 20.1587, 20.1587],
                       [-25.3984, -25.3984, 25.3984, 25.3984],
                       [-11.3137, -22.6274, 11.3137, 22.6274],
                       [-14.2544, -28.5088, 14.2544, 28.5088],
                       [-17.9594, -35.9188, 17.9594, 35.9188]], np.float32)
    for l, base in enumerate(ag.base_anchors):
        # every level is level 0 scaled by the stride ratio (the reference's
        # table lists all five; they are 2^l multiples to its 4 decimals)
        want = torch.tensor(level0 * 2 ** l)
        assert base.allclose(want, rtol=1e-5, atol=2e-4 * 2 ** l), l
        assert np.allclose(O.retina_base_anchors(8 * 2 ** l), want.numpy(),
                           rtol=1e-5, atol=2e-4 * 2 ** l)
        assert np.array_equal(O.retina_base_anchors(8 * 2 ** l), base.numpy())
    assert ag.num_base_anchors == [9, 9, 9, 9, 9]
    sizes = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    flags = ag.valid_flags(sizes, (640, 640), 'cpu')
    assert [int(f.sum()) for f in flags] == [57600, 14400, 3600, 900, 225]
    anchors = ag.grid_anchors(sizes, 'cpu')
    assert len(anchors) == 5
    assert [a.shape[0] for a in anchors] == [h * w * 9 for h, w in sizes]
    for a, b in zip(anchors, O.retina_grid_anchors(sizes)):
        assert np.array_equal(a.numpy(), b)
    # the single-square generator of the GFL configs still reports one anchor
    sq = build_anchor_generator(dict(
        type='AnchorGenerator', ratios=[1.0], octave_base_scale=8,
        scales_per_octave=1, strides=[8, 16, 32, 64, 128]))
    assert sq.num_base_anchors == [1] * 5 and sq.single_square
    assert torch.equal(sq.base_anchors[1],
                       torch.tensor([[-64., -64., 64., 64.]]))
