"""GPU parity (-m gpu) of the inference post-processing (SURVEY.md section 8f
rank 1): ld_get_bboxes through the C ABI and through GFLHead.get_bboxes, against
 (1) the REFERENCE's get_bboxes outputs (tests/golden/infer.npz) and
 (2) the numpy oracle on the same seeded inputs.
Bar: detection count, classes and order exact; coordinates within 1e-3 px,
scores within 1e-6.  (Two detections whose scores differ by < 5e-7 may swap:
the device and torch-CPU sigmoids differ in the last ulp.)"""
import os
import sys

import numpy as np
import pytest
import torch

from ld_amd import synthetic

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'oracle'))

CASES = {c[0]: c for c in synthetic.INFER_CASES}
STRIDES = (8, 16, 32, 64, 128)


def _same(dets, labels, gd, gl, what):
    assert dets.shape == gd.shape, f'{what}: {dets.shape} vs {gd.shape}'
    n = gd.shape[0]
    used = np.zeros(n, dtype=bool)
    for i in range(n):
        ok = False
        for j in (i, i - 1, i + 1):
            if j < 0 or j >= n or used[j]:
                continue
            if j != i and abs(float(gd[j, 4]) - float(gd[i, 4])) > 5e-7:
                continue
            if labels[i] == gl[j] and \
                    np.abs(dets[i, :4] - gd[j, :4]).max() <= 1e-3 and \
                    abs(float(dets[i, 4]) - float(gd[j, 4])) <= 1e-6:
                used[j] = ok = True
                break
        assert ok, f'{what}: detection {i} {dets[i]} label {labels[i]} ' \
                   f'vs {gd[i]} label {gl[i]}'


def _run(case, rescale, dev):
    from ld_amd import lossblock as LB
    cls, reg, metas = synthetic.infer_inputs(case, device=dev)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas] if rescale else None
    res = LB.get_bboxes(cls, reg, STRIDES, shapes, sfs, nms_pre=case[5],
                        score_thr=0.05, iou_thr=0.6, max_per_img=100)
    return [(d.cpu().numpy(), l.cpu().numpy()) for d, l in res]


@pytest.mark.parametrize('rescale', [False, True], ids=['r0', 'r1'])
@pytest.mark.parametrize('name', list(CASES))
def test_get_bboxes_vs_reference_golden(golden, name, rescale):
    dev = torch.device('cuda:0')
    g = golden['infer']
    res = _run(CASES[name], rescale, dev)
    for i, (dets, labels) in enumerate(res):
        tag = f'{name}_r{int(rescale)}'
        _same(dets, labels, g[f'{tag}_bboxes_{i}'], g[f'{tag}_labels_{i}'],
              f'{tag} image {i}')
        assert np.all(np.diff(dets[:, 4]) <= 0)


@pytest.mark.parametrize('name', ['small', 'small_topk', 'c2'])
def test_get_bboxes_vs_oracle(name):
    import ld_oracle as O
    dev = torch.device('cuda:0')
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs(case)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas]
    ref = O.get_bboxes([c.numpy() for c in cls], [r.numpy() for r in reg],
                       shapes, sfs, nms_pre=case[5], rescale=True)
    res = _run(case, True, dev)
    for i, ((dets, labels), (rd, rl)) in enumerate(zip(res, ref)):
        _same(dets, labels, rd, rl, f'{name} image {i}')


def test_head_api_and_edge_cases(golden):
    """GFLHead.get_bboxes(cls_scores, bbox_preds, img_metas, cfg, rescale)
    (anchor_head.py:497-589 signature); a threshold nothing passes -> empty
    (0, 5) / (0,) results; max_per_img truncation keeps the best."""
    from ld_amd import model_zoo
    from ld_amd.registry import build_detector
    dev = torch.device('cuda:0')
    det = build_detector(model_zoo.gfl_detector(18)).to(dev)
    head = det.bbox_head
    case = CASES['small']
    cls, reg, metas = synthetic.infer_inputs(case, device=dev)
    cfg = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05,
               nms=dict(type='nms', iou_threshold=0.6), max_per_img=100)
    g = golden['infer']
    res = head.get_bboxes(cls, reg, metas, cfg=cfg, rescale=True)
    for i, (d, l) in enumerate(res):
        assert d.dtype == torch.float32 and l.dtype == torch.int64
        _same(d.cpu().numpy(), l.cpu().numpy(), g[f'small_r1_bboxes_{i}'],
              g[f'small_r1_labels_{i}'], f'head api image {i}')
    cfg10 = dict(cfg, max_per_img=10)
    res10 = head.get_bboxes(cls, reg, metas, cfg=cfg10, rescale=True)
    for (d, l), (d10, l10) in zip(res, res10):
        assert d10.shape == (10, 5)
        assert torch.equal(d10, d[:10]) and torch.equal(l10, l[:10])
    cfg_none = dict(cfg, score_thr=0.999999)
    for d, l in head.get_bboxes(cls, reg, metas, cfg=cfg_none):
        assert tuple(d.shape) == (0, 5) and tuple(l.shape) == (0, )
    with pytest.raises(NotImplementedError):
        head.get_bboxes(cls, reg, metas, cfg=dict(
            cfg, nms=dict(type='soft_nms', iou_threshold=0.6)))


def test_detector_simple_test():
    """SingleStageDetector.forward(return_loss=False) -> forward_test ->
    simple_test (single_stage.py:98-129, base.py:120-183): per image a list of
    num_classes (k_c, 5) float32 arrays, equal to the head-level call."""
    from ld_amd import model_zoo, synthetic as S
    dev = torch.device('cuda:0')
    det = model_zoo.build_seeded_ld_detector(18, 18, dev)
    det.eval()
    det.bbox_head.test_cfg['score_thr'] = 0.001  # seeded weights score low
    batch = S.synthetic_batch(2, (120, 150), (128, 160), [2, 3], 7)
    img = batch['img'].to(dev)
    metas = batch['img_metas']
    for m, sf in zip(metas, (1.0, 1.25)):
        m['scale_factor'] = np.array([sf] * 4, dtype=np.float32)
    res = det(img=[img], img_metas=[metas], return_loss=False, rescale=True)
    assert len(res) == 2
    with torch.no_grad():
        outs = det.bbox_head(det.extract_feat(img))
        ref = det.bbox_head.get_bboxes(*outs, metas, rescale=True)
    total = 0
    for per_cls, (db, dl) in zip(res, ref):
        assert len(per_cls) == 80
        db, dl = db.cpu().numpy(), dl.cpu().numpy()
        for c, arr in enumerate(per_cls):
            assert arr.dtype == np.float32 and arr.shape[1] == 5
            assert np.array_equal(arr, db[dl == c])
        assert sum(a.shape[0] for a in per_cls) == db.shape[0] <= 100
        total += db.shape[0]
    assert total > 0
    with pytest.raises(TypeError):
        det(img=img, img_metas=[metas], return_loss=False)


def test_global_sort_path(golden, monkeypatch):
    """LD_INFER_SORT=global: the plain global-memory bitonic sorts (the
    fallback of the radix-select + LDS-sort fast path) give the same result."""
    monkeypatch.setenv('LD_INFER_SORT', 'global')
    dev = torch.device('cuda:0')
    g = golden['infer']
    for i, (dets, labels) in enumerate(_run(CASES['c2'], False, dev)):
        _same(dets, labels, g[f'c2_r0_bboxes_{i}'], g[f'c2_r0_labels_{i}'],
              f'global path image {i}')


def test_fallback_when_best_candidates_run_out(monkeypatch):
    """With the fast path's candidate window shrunk to 256 (LD_INFER_LIMIT, a
    test hook) and max_per_img = 1024 the window runs out, so the call must
    fall back to the fully sorted list (6.8 k candidates) -- and still equal
    the oracle (iou_threshold 0: every overlap within a class suppresses)."""
    import ld_oracle as O
    from ld_amd import lossblock as LB
    monkeypatch.setenv('LD_INFER_LIMIT', '256')
    dev = torch.device('cuda:0')
    case = CASES['c2']
    cls, reg, metas = synthetic.infer_inputs(case)
    shapes = [m['img_shape'] for m in metas]
    ref = O.get_bboxes([c.numpy() for c in cls], [r.numpy() for r in reg],
                       shapes, None, nms_pre=case[5], iou_thr=0.0,
                       max_per_img=1024, rescale=False)
    res = LB.get_bboxes([c.to(dev) for c in cls], [r.to(dev) for r in reg],
                        STRIDES, shapes, None, nms_pre=case[5], score_thr=0.05,
                        iou_thr=0.0, max_per_img=1024)
    for i, ((d, l), (rd, rl)) in enumerate(zip(res, ref)):
        assert rd.shape[0] > 256  # the premise: more keeps than the window
        _same(d.cpu().numpy(), l.cpu().numpy(), rd, rl, f'fallback image {i}')
    with pytest.raises(Exception):  # negative thresholds are rejected
        LB.get_bboxes([c.to(dev) for c in cls], [r.to(dev) for r in reg],
                      STRIDES, shapes, None, nms_pre=case[5], iou_thr=-1.0)


VCASES = {c[0]: c for c in synthetic.VOTING_CASES}


@pytest.mark.parametrize('rescale', [False, True], ids=['r0', 'r1'])
@pytest.mark.parametrize('thr', [0.6, 0.85])
@pytest.mark.parametrize('name', list(VCASES))
def test_voting_nms_vs_reference_golden(golden, name, thr, rescale):
    """ld_get_bboxes_voting against the REFERENCE's own
    multiclass_nms(type='voting_cluster_diounms') outputs (pure torch in the
    reference: this NMS variant is parity-pinned end to end)."""
    from ld_amd import lossblock as LB
    dev = torch.device('cuda:0')
    g = golden['infer_voting']
    case = VCASES[name]
    cls, reg, metas = synthetic.voting_inputs(case, device=dev)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas] if rescale else None
    res = LB.get_bboxes(cls, reg, STRIDES, shapes, sfs, nms_pre=case[5],
                        score_thr=0.05, iou_thr=thr, max_per_img=100,
                        voting=True)
    for i, (d, l) in enumerate(res):
        tag = f'{name}_t{int(thr * 100)}_r{int(rescale)}'
        dets, labels = d.cpu().numpy(), l.cpu().numpy()
        gd, gl = g[f'{tag}_bboxes_{i}'], g[f'{tag}_labels_{i}']
        assert dets.shape == gd.shape
        assert np.array_equal(labels, gl), tag
        np.testing.assert_allclose(dets[:, :4], gd[:, :4], atol=2e-3, rtol=0,
                                   err_msg=tag)
        np.testing.assert_allclose(dets[:, 4], gd[:, 4], atol=1e-6, rtol=0)


def test_voting_head_api():
    """cfg.nms.type = 'voting_cluster_diounms' through GFLHead.get_bboxes."""
    from ld_amd import model_zoo
    from ld_amd.registry import build_head
    dev = torch.device('cuda:0')
    case = synthetic.VOTING_CASES[1]
    cls, reg, metas = synthetic.voting_inputs(case, device=dev)
    cfg = dict(model_zoo.gfl_detector(50)['bbox_head'])
    head = build_head(cfg).to(dev)
    tc = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05,
              nms=dict(type='voting_cluster_diounms', iou_threshold=0.6),
              max_per_img=100)
    voted = head.get_bboxes(cls, reg, metas, cfg=tc)
    tc['nms'] = dict(type='nms', iou_threshold=0.6)
    plain = head.get_bboxes(cls, reg, metas, cfg=tc)
    assert voted[0][0].shape == (100, 5)
    # same top-scoring class, different (averaged) coordinates
    assert int(voted[0][1][0]) == int(plain[0][1][0])
    assert float((voted[0][0][0, :4] - plain[0][0][0, :4]).abs().max()) > 1e-2
    with pytest.raises(NotImplementedError):
        tc['nms'] = dict(type='soft_nms', iou_threshold=0.6)
        head.get_bboxes(cls, reg, metas, cfg=tc)


# ---------------------------------------------------------------------------
# GFocalHead.get_bboxes (GFLv2 / LDv2Head: BASELINE config 5's student)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('rescale', [False, True], ids=['r0', 'r1'])
@pytest.mark.parametrize('name', synthetic.INFER_V2_CASES)
def test_gfocal_get_bboxes_vs_reference_golden(golden, name, rescale):
    """LD_INFER_PROB: no sigmoid, 81 score channels; against the reference's
    GFocalHead.get_bboxes outputs (tests/golden/infer_v2.npz)."""
    from ld_amd import lossblock as LB
    dev = torch.device('cuda:0')
    g = golden['infer_v2']
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs_prob(case, device=dev)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas] if rescale else None
    res = LB.get_bboxes(cls, reg, STRIDES, shapes, sfs, nms_pre=case[5],
                        score_thr=0.05, iou_thr=0.6, max_per_img=100,
                        prob=True)
    seen80 = False
    for i, (d, l) in enumerate(res):
        tag = f'{name}_r{int(rescale)}'
        dets, labels = d.cpu().numpy(), l.cpu().numpy()
        _same(dets, labels, g[f'{tag}_bboxes_{i}'], g[f'{tag}_labels_{i}'],
              f'v2 {tag} image {i}')
        seen80 = seen80 or bool((labels == 80).any())
    assert seen80


def test_gfocal_head_api_and_detector_simple_test():
    """GFocalHead.get_bboxes(cls_scores, bbox_preds, cls_feat, img_metas, ...)
    (the reference's four-positional signature) and the LDv2 detector's
    simple_test through it."""
    from ld_amd import model_zoo, synthetic as S
    from ld_amd.registry import build_detector
    dev = torch.device('cuda:0')
    det = build_detector(model_zoo.ldv2_detector(18, 18))
    det.load_state_dict(S.seeded_state_dict(det.state_dict(), seed=1))
    det.to(dev).eval()
    det.bbox_head.test_cfg['score_thr'] = 0.001
    batch = S.synthetic_batch(2, (120, 150), (128, 160), [2, 3], 7)
    img = batch['img'].to(dev)
    metas = batch['img_metas']
    for m, sf in zip(metas, (1.0, 1.25)):
        m['scale_factor'] = np.array([sf] * 4, dtype=np.float32)
    res = det(img=[img], img_metas=[metas], return_loss=False, rescale=True)
    with torch.no_grad():
        outs = det.bbox_head(det.extract_feat(img))
        assert len(outs) == 3
        ref = det.bbox_head.get_bboxes(*outs, metas, rescale=True)
        # probabilities in, so the plain-GFL entry on logit(p) must agree
        cls_p = [c.clamp(1e-6, 1 - 1e-6) for c in outs[0]]
    total = 0
    for per_cls, (db, dl) in zip(res, ref):
        assert len(per_cls) == 80
        db, dl = db.cpu().numpy(), dl.cpu().numpy()
        for c, arr in enumerate(per_cls):
            assert np.array_equal(arr, db[dl == c])
        total += db.shape[0]
        assert db.shape[0] <= 100 and np.all(np.diff(db[:, 4]) <= 0)
        assert db[:, 4].max() <= 1.0 and db[:, 4].min() > 0.001
    assert total > 0
    assert cls_p[0].shape[1] == 81


# ---------------------------------------------------------------------------
# ATSSGFLHead / FCOSGFLHead get_bboxes (centerness factor, FCOS points)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('kind', ['atss', 'fcos'])
@pytest.mark.parametrize('rescale', [False, True], ids=['r0', 'r1'])
@pytest.mark.parametrize('name', synthetic.INFER_V2_CASES)
def test_centerness_get_bboxes_vs_reference_golden(golden, name, rescale, kind):
    from ld_amd import lossblock as LB
    dev = torch.device('cuda:0')
    g = golden['infer_ctr']
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs(case, device=dev)
    ctr = synthetic.synthetic_centerness(
        len(metas), synthetic.level_shapes(case[1]), seed=case[4], device=dev)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas] if rescale else None
    res = LB.get_bboxes(cls, reg, STRIDES, shapes, sfs, nms_pre=case[5],
                        score_thr=0.05, iou_thr=0.6, max_per_img=100,
                        centernesses=ctr, points=kind == 'fcos')
    for i, (d, l) in enumerate(res):
        tag = f'{kind}_{name}_r{int(rescale)}'
        _same(d.cpu().numpy(), l.cpu().numpy(), g[f'{tag}_bboxes_{i}'],
              g[f'{tag}_labels_{i}'], f'{tag} image {i}')


def test_atss_fcos_head_get_bboxes_api():
    """ATSSGFLHead / FCOSGFLHead.get_bboxes(cls, reg, centernesses, metas):
    equal to the C-ABI-level call; the FCOS boxes differ from the ATSS ones by
    the half-stride point offset."""
    from ld_amd import lossblock as LB
    from ld_amd.config import ConfigDict
    from ld_amd.registry import build_head
    from ld_amd import model_zoo
    dev = torch.device('cuda:0')
    case = CASES['small']
    cls, reg, metas = synthetic.infer_inputs(case, device=dev)
    ctr = synthetic.synthetic_centerness(
        len(metas), synthetic.level_shapes(case[1]), seed=case[4], device=dev)
    test_cfg = ConfigDict.wrap(dict(
        nms_pre=1000, min_bbox_size=0, score_thr=0.05,
        nms=dict(type='nms', iou_threshold=0.6), max_per_img=100))
    out = {}
    for kind, hc in (('atss', model_zoo.atss_gfl_detector(50)['bbox_head']),
                     ('fcos', model_zoo.fcos_gfl_detector(50)['bbox_head'])):
        head = build_head(dict(hc, test_cfg=test_cfg)).to(dev)
        got = head.get_bboxes(cls, reg, ctr, metas, rescale=True)
        want = LB.get_bboxes(cls, reg, STRIDES,
                             [m['img_shape'] for m in metas],
                             [m['scale_factor'] for m in metas], nms_pre=1000,
                             score_thr=0.05, iou_thr=0.6, max_per_img=100,
                             centernesses=ctr, points=kind == 'fcos')
        for (a, b), (c, d) in zip(got, want):
            assert torch.equal(a, c) and torch.equal(b, d)
        out[kind] = got
        with pytest.raises(NotImplementedError):
            head.get_bboxes(cls, reg, ctr, metas, cfg=ConfigDict.wrap(dict(
                test_cfg, nms=dict(type='voting_cluster_diounms',
                                   iou_threshold=0.6))))
    assert not torch.equal(out['atss'][0][0], out['fcos'][0][0])


# ---------------------------------------------------------------------------
# RetinaGFLHead get_bboxes (9 anchors per cell)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('rescale', [False, True], ids=['r0', 'r1'])
@pytest.mark.parametrize('name', synthetic.INFER_V2_CASES)
def test_retina_get_bboxes_vs_reference_golden(golden, name, rescale):
    from ld_amd import lossblock as LB
    dev = torch.device('cuda:0')
    g = golden['infer_retina']
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs_retina(case, device=dev)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas] if rescale else None
    res = LB.get_bboxes(cls, reg, STRIDES, shapes, sfs, nms_pre=case[5],
                        score_thr=0.05, iou_thr=0.6, max_per_img=100,
                        num_base=9)
    for i, (d, l) in enumerate(res):
        tag = f'{name}_r{int(rescale)}'
        _same(d.cpu().numpy(), l.cpu().numpy(), g[f'{tag}_bboxes_{i}'],
              g[f'{tag}_labels_{i}'], f'retina {tag} image {i}')


def test_retina_detector_simple_test():
    """The LD-Retina detector's simple_test through RetinaGFLHead.get_bboxes
    (level-sliced views of the (N, 720, P) head output)."""
    from ld_amd import model_zoo, synthetic as S
    from ld_amd.registry import build_detector
    dev = torch.device('cuda:0')
    det = build_detector(model_zoo.ld_retina_detector(18, 18))
    det.load_state_dict(S.seeded_state_dict(det.state_dict(), seed=1))
    det.to(dev).eval()
    det.bbox_head.test_cfg['score_thr'] = 0.001
    batch = S.synthetic_batch(2, (120, 150), (128, 160), [2, 3], 7)
    img = batch['img'].to(dev)
    metas = batch['img_metas']
    for m, sf in zip(metas, (1.0, 1.25)):
        m['scale_factor'] = np.array([sf] * 4, dtype=np.float32)
    res = det(img=[img], img_metas=[metas], return_loss=False, rescale=True)
    with torch.no_grad():
        outs = det.bbox_head(det.extract_feat(img))
        ref = det.bbox_head.get_bboxes(*outs, metas, rescale=True)
    for per_cls, (db, dl) in zip(res, ref):
        db, dl = db.cpu().numpy(), dl.cpu().numpy()
        assert len(per_cls) == 80 and db.shape[0] <= 100
        for c, arr in enumerate(per_cls):
            assert np.array_equal(arr, db[dl == c])
        assert np.all(np.diff(db[:, 4]) <= 0)


# ---------------------------------------------------------------------------
# get_bboxes(with_nms=False): the pre-NMS stage (ld_get_bboxes_pre_nms)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['small', 'small_topk', 'c2'])
def test_with_nms_false_vs_reference_golden(golden, name):
    """GFLHead.get_bboxes(..., with_nms=False) -> per image (mlvl_bboxes,
    mlvl_scores with the zero background column) against the reference's own
    pre-NMS outputs (tests/golden/infer.npz: stored arrays for the small cases,
    counts / sums / per-row max scores at the C2 size).  Rows inside a sorted
    level may swap where two max-scores tie to 1e-7 (torch.topk's order)."""
    from ld_amd.config import ConfigDict
    from ld_amd.registry import build_head
    from ld_amd import model_zoo
    dev = torch.device('cuda:0')
    g = golden['infer']
    case = CASES[name]
    cls, reg, metas = synthetic.infer_inputs(case, device=dev)
    hc = dict(model_zoo.gfl_detector(50)['bbox_head'])
    head = build_head(dict(hc, test_cfg=ConfigDict.wrap(dict(
        nms_pre=case[5], min_bbox_size=0, score_thr=0.05,
        nms=dict(type='nms', iou_threshold=0.6), max_per_img=100)))).to(dev)
    res = head.get_bboxes(cls, reg, metas, rescale=False, with_nms=False)
    for i, (bb, sc) in enumerate(res):
        bb, sc = bb.cpu().numpy(), sc.cpu().numpy()
        assert bb.shape[0] == int(g[f'{name}_pre_count_{i}'])
        assert sc.shape == (bb.shape[0], 81) and not sc[:, 80].any()
        sc = sc[:, :80]
        assert int((sc > np.float32(0.05)).sum()) == \
            int(g[f'{name}_candidates_{i}'])
        if case[8]:
            gb, gs = g[f'{name}_pre_bboxes_{i}'], g[f'{name}_pre_scores_{i}']
            # match rows up to swaps between equal-key neighbours
            key_d, key_g = sc.max(1), gs.max(1)
            np.testing.assert_allclose(key_d, key_g, atol=1e-6, rtol=0)
            bad = (np.abs(bb - gb).max(1) > 1e-3) | \
                (np.abs(sc - gs).max(1) > 1e-6)
            for r in np.nonzero(bad)[0]:
                nb = [q for q in (r - 1, r + 1) if 0 <= q < len(bad) and
                      abs(float(key_g[q]) - float(key_g[r])) <= 5e-7]
                assert any(np.abs(bb[r] - gb[q]).max() <= 1e-3 and
                           np.abs(sc[r] - gs[q]).max() <= 1e-6 for q in nb), r
        else:
            np.testing.assert_allclose(sc.max(1), g[f'{name}_pre_maxscore_{i}'],
                                       atol=1e-6, rtol=0)
            np.testing.assert_allclose(
                float(sc.astype(np.float64).sum()),
                float(g[f'{name}_pre_scores_sum_{i}']), rtol=1e-5)
            np.testing.assert_allclose(
                float(bb.astype(np.float64).sum()),
                float(g[f'{name}_pre_bboxes_sum_{i}']), rtol=1e-5)
    # the centerness heads return the factors as a third array
    ctr = synthetic.synthetic_centerness(
        len(metas), synthetic.level_shapes(case[1]), seed=case[4], device=dev)
    ah = build_head(dict(model_zoo.atss_gfl_detector(50)['bbox_head'],
                         test_cfg=head.test_cfg)).to(dev)
    out = ah.get_bboxes(cls, reg, ctr, metas, with_nms=False)
    assert len(out[0]) == 3 and out[0][2].shape == (out[0][0].shape[0], )
    assert float(out[0][2].min()) > 0 and float(out[0][2].max()) < 1
