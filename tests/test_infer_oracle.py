"""Inference post-processing: the oracle's
restatement of GFLHead.get_bboxes -- sigmoid scores, Integral * stride, per-level
top-nms_pre, distance2bbox + clamp, multiclass_nms / batched_nms -- against the
outputs of the REFERENCE get_bboxes executed under the shim
(tests/golden/infer.npz, oracle/gen_golden.py --only infer).  CPU only: this
pins the checker the HIP path (tests/test_gpu_infer.py) is held to (labels,
order and counts exact; coordinates within 1e-3 px, scores 1e-6)."""
import sys

import numpy as np
import pytest

from ld_amd import synthetic

sys.path.insert(0, __import__('os').path.join(
    __import__('os').path.dirname(__import__('os').path.dirname(
        __import__('os').path.abspath(__file__))), 'oracle'))
import ld_oracle as O  # noqa: E402

CASES = {c[0]: c for c in synthetic.INFER_CASES}


def _inputs(case):
    cls, reg, metas = synthetic.infer_inputs(case)
    return ([c.numpy() for c in cls], [r.numpy() for r in reg],
            [m['img_shape'] for m in metas],
            [m['scale_factor'] for m in metas])


@pytest.mark.parametrize('name', list(CASES))
def test_get_bboxes_vs_reference(golden, name):
    g = golden['infer']
    case = CASES[name]
    cls, reg, shapes, sfs = _inputs(case)
    for rs in (0, 1):
        res = O.get_bboxes(cls, reg, shapes, sfs, nms_pre=case[5],
                           rescale=bool(rs))
        for i, (dets, labels) in enumerate(res):
            gd = g[f'{name}_r{rs}_bboxes_{i}']
            gl = g[f'{name}_r{rs}_labels_{i}']
            assert dets.shape == gd.shape
            assert np.array_equal(labels, gl)  # class AND order
            np.testing.assert_allclose(dets[:, :4], gd[:, :4], atol=1e-3,
                                       rtol=0)
            np.testing.assert_allclose(dets[:, 4], gd[:, 4], atol=1e-6,
                                       rtol=0)
            # sorted by score, at most max_per_img, inside the image
            assert dets.shape[0] <= 100
            assert np.all(np.diff(dets[:, 4]) <= 0)


@pytest.mark.parametrize('name', list(CASES))
def test_pre_nms_stage_vs_reference(golden, name):
    g = golden['infer']
    case = CASES[name]
    cls, reg, shapes, sfs = _inputs(case)
    pre = O.get_bboxes_pre_nms(cls, reg, shapes, case[5])
    for i, (bb, sc) in enumerate(pre):
        assert bb.shape[0] == int(g[f'{name}_pre_count_{i}'])
        assert int((sc > np.float32(0.05)).sum()) == \
            int(g[f'{name}_candidates_{i}'])
        H, W = shapes[i][:2]
        assert bb[:, 0::2].min() >= 0 and bb[:, 0::2].max() <= W
        assert bb[:, 1::2].min() >= 0 and bb[:, 1::2].max() <= H
        if case[8]:
            np.testing.assert_allclose(bb, g[f'{name}_pre_bboxes_{i}'],
                                       atol=1e-3, rtol=0)
            np.testing.assert_allclose(sc, g[f'{name}_pre_scores_{i}'],
                                       atol=1e-6, rtol=0)
        else:
            np.testing.assert_allclose(sc.max(1), g[f'{name}_pre_maxscore_{i}'],
                                       atol=1e-6, rtol=0)
            np.testing.assert_allclose(float(bb.astype(np.float64).sum()),
                                       float(g[f'{name}_pre_bboxes_sum_{i}']),
                                       rtol=1e-6)
            np.testing.assert_allclose(float(sc.astype(np.float64).sum()),
                                       float(g[f'{name}_pre_scores_sum_{i}']),
                                       rtol=1e-6)


def test_cases_cover_both_batched_nms_branches(golden):
    g = golden['infer']
    cands = {n: [int(g[f'{n}_candidates_{i}']) for i in range(2)]
             for n in CASES}
    assert max(cands['c2']) < 10000 <= min(cands['c2_dense'])
    assert all(int(g['small_topk_pre_count_0']) < int(g['small_pre_count_0'])
               for _ in [0])


def test_nms_known_answers():
    """Hand-checkable cases of the greedy NMS / class-offset semantics."""
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30]], np.float32)
    s = np.array([[0.9, 0.1], [0.8, 0.2], [0.3, 0.7]], np.float32)
    dets, labels = O.multiclass_nms(b, s, 0.05, 0.6, 100)
    # IoU([0,0,10,10],[1,1,11,11]) = 81/119 = 0.68 > 0.6 within a class only
    assert labels.tolist() == [0, 1, 0, 1]
    np.testing.assert_allclose(dets[:, 4], [0.9, 0.7, 0.3, 0.2], atol=1e-7)
    # exactly-at-threshold is kept (suppression is strict >)
    b2 = np.array([[0, 0, 10, 10], [0, 0, 10, 6]], np.float32)  # IoU 0.6
    keep = O.nms_greedy(b2, np.array([0.9, 0.8], np.float32), 0.6)
    assert keep.tolist() == [0, 1]
    # ties: lower index first
    keep = O.nms_greedy(np.array([[0, 0, 1, 1], [5, 5, 6, 6]], np.float32),
                        np.array([0.5, 0.5], np.float32), 0.6)
    assert keep.tolist() == [0, 1]
    # nothing above the threshold
    d, l = O.multiclass_nms(b, s * 0.01, 0.05, 0.6, 100)
    assert d.shape == (0, 5) and l.shape == (0, )
    # max_num truncation keeps the best
    d, l = O.multiclass_nms(b, s, 0.05, 0.6, 2)
    assert l.tolist() == [0, 1] and d.shape == (2, 5)


VCASES = {c[0]: c for c in synthetic.VOTING_CASES}


@pytest.mark.parametrize('thr', [0.6, 0.85])
@pytest.mark.parametrize('name', list(VCASES))
def test_voting_nms_vs_reference(golden, name, thr):
    """Score-voting Cluster-DIoU-NMS (nms type 'voting_cluster_diounms'): the
    reference branch is pure torch (bbox_nms.py:141-176), so these goldens PIN
    it end to end."""
    g = golden['infer_voting']
    case = VCASES[name]
    cls, reg, metas = synthetic.voting_inputs(case)
    cls, reg = [c.numpy() for c in cls], [r.numpy() for r in reg]
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas]
    moved = 0.0
    for rs in (0, 1):
        res = O.get_bboxes(cls, reg, shapes, sfs, nms_pre=case[5], iou_thr=thr,
                           rescale=bool(rs), voting=True)
        plain = O.get_bboxes(cls, reg, shapes, sfs, nms_pre=case[5],
                             iou_thr=thr, rescale=bool(rs))
        for i, (dets, labels) in enumerate(res):
            tag = f'{name}_t{int(thr * 100)}_r{rs}'
            gd, gl = g[f'{tag}_bboxes_{i}'], g[f'{tag}_labels_{i}']
            assert dets.shape == gd.shape
            assert np.array_equal(labels, gl)
            np.testing.assert_allclose(dets[:, :4], gd[:, :4], atol=2e-3,
                                       rtol=0)
            np.testing.assert_allclose(dets[:, 4], gd[:, 4], atol=1e-6, rtol=0)
            # the voted box differs from the un-voted candidate it came from
            same = min(len(dets), len(plain[i][0]))
            if same and np.array_equal(labels[:1], plain[i][1][:1]):
                moved = max(moved, float(np.abs(
                    dets[0, :4] - plain[i][0][0, :4]).max()))
    if case[6]:
        assert moved > 1e-2, 'clustered case: voting must move the top box'


@pytest.mark.parametrize('name', synthetic.INFER_V2_CASES)
def test_gfocal_get_bboxes_vs_reference(golden, name):
    """GFocalHead.get_bboxes (prob=True: no sigmoid, 81 score channels with
    the background column an ordinary class) against the reference's outputs
    (tests/golden/infer_v2.npz)."""
    g = golden['infer_v2']
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs_prob(case)
    cls, reg = [c.numpy() for c in cls], [r.numpy() for r in reg]
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas]
    seen80 = False
    for rs in (0, 1):
        res = O.get_bboxes(cls, reg, shapes, sfs, nms_pre=case[5],
                           rescale=bool(rs), prob=True)
        for i, (dets, labels) in enumerate(res):
            gd = g[f'{name}_r{rs}_bboxes_{i}']
            gl = g[f'{name}_r{rs}_labels_{i}']
            assert dets.shape == gd.shape
            assert np.array_equal(labels, gl)
            np.testing.assert_allclose(dets[:, :4], gd[:, :4], atol=1e-3,
                                       rtol=0)
            np.testing.assert_allclose(dets[:, 4], gd[:, 4], atol=1e-6,
                                       rtol=0)
            seen80 = seen80 or bool((labels == 80).any())
    assert seen80  # the 81st channel really takes part


@pytest.mark.parametrize('kind', ['atss', 'fcos'])
@pytest.mark.parametrize('name', synthetic.INFER_V2_CASES)
def test_centerness_get_bboxes_vs_reference(golden, name, kind):
    """ATSSGFLHead / FCOSGFLHead get_bboxes: top-k by score x centerness, the
    factor applied to the scores after the threshold test, FCOS points; against
    the reference's outputs (tests/golden/infer_ctr.npz)."""
    g = golden['infer_ctr']
    case = CASES[name]
    cls, reg, shapes, sfs = _inputs(case)
    ctr = [c.numpy() for c in synthetic.synthetic_centerness(
        len(shapes), synthetic.level_shapes(case[1]), seed=case[4])]
    below = False
    for rs in (0, 1):
        res = O.get_bboxes(cls, reg, shapes, sfs, nms_pre=case[5],
                           rescale=bool(rs), centernesses=ctr,
                           points=kind == 'fcos')
        for i, (dets, labels) in enumerate(res):
            gd = g[f'{kind}_{name}_r{rs}_bboxes_{i}']
            gl = g[f'{kind}_{name}_r{rs}_labels_{i}']
            assert dets.shape == gd.shape
            assert np.array_equal(labels, gl)
            np.testing.assert_allclose(dets[:, :4], gd[:, :4], atol=1e-3,
                                       rtol=0)
            np.testing.assert_allclose(dets[:, 4], gd[:, 4], atol=1e-6,
                                       rtol=0)
            below = below or bool((dets[:, 4] < 0.05).any())
    if name == 'small_topk':
        assert below  # a final score may sit below score_thr: factor applied after


@pytest.mark.parametrize('name', synthetic.INFER_V2_CASES)
def test_retina_get_bboxes_vs_reference(golden, name):
    """RetinaGFLHead.get_bboxes: 9 anchors per cell, top-k per level over all
    (cell, anchor) rows; against the reference's outputs
    (tests/golden/infer_retina.npz)."""
    g = golden['infer_retina']
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs_retina(case)
    cls, reg = [c.numpy() for c in cls], [r.numpy() for r in reg]
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas]
    for rs in (0, 1):
        res = O.get_bboxes(cls, reg, shapes, sfs, nms_pre=case[5],
                           rescale=bool(rs), num_base=9)
        for i, (dets, labels) in enumerate(res):
            gd = g[f'{name}_r{rs}_bboxes_{i}']
            gl = g[f'{name}_r{rs}_labels_{i}']
            assert dets.shape == gd.shape
            assert np.array_equal(labels, gl)
            np.testing.assert_allclose(dets[:, :4], gd[:, :4], atol=1e-3,
                                       rtol=0)
            np.testing.assert_allclose(dets[:, 4], gd[:, 4], atol=1e-6,
                                       rtol=0)
