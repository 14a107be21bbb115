"""LDFCOSHead: the numpy restatement of the FCOS
point targets and of LDFCOSHead.loss (oracle/ld_oracle.py; ld_fcos_head.py
over fcos_gfl_head.py) against what the REFERENCE produced
(tests/golden/lossblock_fcos.npz, oracle/gen_golden.py gen_lossblock_fcos)."""
import os
import sys

import numpy as np
import pytest

from ld_amd import synthetic

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'oracle'))
import ld_oracle as O  # noqa: E402
from test_oracle_atss import check_grads, inputs  # noqa: E402

CASES = ['small', 'small_crowd', 'c2', 'c2_crowd']


def reference_targets(g, name, sizes, N):
    """(labels (N, A), bbox_targets (N, A, 4)) from the per-level, image-
    concatenated golden arrays."""
    labs, bts = [], []
    for l, (h, w) in enumerate(sizes):
        labs.append(g[f'{name}_labels_{l}'].reshape(N, h * w))
        bts.append(g[f'{name}_bbox_targets_{l}'].reshape(N, h * w, 4))
    return np.concatenate(labs, 1), np.concatenate(bts, 1)


@pytest.mark.parametrize('name', CASES)
def test_fcos_targets_bit_exact(golden, name):
    g = golden['lossblock_fcos']
    batch, sizes, hi = inputs(g, name)
    t = O.fcos_targets(sizes, [b.numpy() for b in batch['gt_bboxes']],
                       [l.numpy() for l in batch['gt_labels']])
    rl, rb = reference_targets(g, name, sizes, len(batch['gt_bboxes']))
    np.testing.assert_array_equal(t['labels'], rl)
    assigned = rl < 80
    np.testing.assert_array_equal(t['bbox_targets'][assigned], rb[assigned])
    assert (rl == 81).sum() > 0 and assigned.sum() > 0


@pytest.mark.parametrize('name', CASES)
def test_fcos_lossblock_vs_reference(golden, name):
    g = golden['lossblock_fcos']
    batch, sizes, hi = inputs(g, name)
    hi = {k: [t.numpy() for t in v] for k, v in hi.items()}
    t = O.fcos_targets(sizes, [b.numpy() for b in batch['gt_bboxes']],
                       [l.numpy() for l in batch['gt_labels']])
    out = O.ld_fcos_loss_block(hi['cls'], hi['reg'], hi['ctr'], hi['t_cls'],
                               hi['t_reg'], t)
    np.testing.assert_allclose(out['losses'], g[name + '_losses'], rtol=3e-5,
                               atol=3e-6)
    check_grads(g, name, out['grads'], 2e-4, 3e-8)
