"""Deterministic synthetic inputs for the LD train step.

There is no dataset and no checkpoint on the GPU box, so the bench, the smoke
test and the parity fixtures all use:

* images: ~N(0,1) (normalised-image statistics), columns past ``img_shape``
  zero-filled exactly as ``Pad(size_divisor=32)`` would
  (reference: mmdet/datasets/pipelines/transforms.py:481-509);
* GT boxes: centre uniform in the image, ``w, h = 16 + 584 u^3`` (small-skewed, like a log-uniform law)
  clipped to the image, non-integer xyxy fp32 (avoids ATSS distance ties),
  labels ``randint(0, 80)``;
* weights: :func:`seeded_state_dict` -- every tensor of a ``state_dict`` is
  filled from a CPU ``torch.Generator`` whose seed is derived from the *key
  name*, so the reference model (under ``oracle/ref_shim``) and the HIP model
  receive bit-identical parameters as long as their key names agree (which is
  itself part of the drop-in contract, SURVEY.md section 5 "checkpoint").

Everything is generated on the CPU generator (bit-reproducible across
machines for one torch build) and moved to the device afterwards.
"""
import math
import zlib
from collections import OrderedDict

import torch

__all__ = ['synthetic_batch', 'seeded_state_dict', 'synthetic_head_inputs']


def _gen(seed):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    return g


def _normal(shape, gen):
    """~N(0, 1) as a scaled sum of three uniforms.  Only IEEE-exact ops
    (rand, +, *) are used anywhere in this module: torch.randn / exp / log go
    through vendor-specific SIMD math and differ by an ulp between the build
    container's Xeon and the GPU box's EPYC, which would make 'identical
    inputs' not identical."""
    u = torch.rand(shape, generator=gen) + torch.rand(shape, generator=gen) \
        + torch.rand(shape, generator=gen)
    return (u - 1.5) * 2.0


def synthetic_boxes(num_gt, img_h, img_w, gen, min_size=16.0, max_size=600.0):
    """(num_gt, 4) xyxy fp32 boxes, (num_gt,) int64 labels."""
    cx = torch.rand(num_gt, generator=gen) * img_w
    cy = torch.rand(num_gt, generator=gen) * img_h
    # sizes skewed to small boxes (u^3 stands in for a log-uniform law)
    uw = torch.rand(num_gt, generator=gen)
    uh = torch.rand(num_gt, generator=gen)
    w = min_size + (max_size - min_size) * (uw * uw * uw)
    h = min_size + (max_size - min_size) * (uh * uh * uh)
    x1 = (cx - w / 2).clamp(0.0, img_w - 2.0)
    y1 = (cy - h / 2).clamp(0.0, img_h - 2.0)
    x2 = torch.maximum((cx + w / 2).clamp(0.0, float(img_w)), x1 + 1.7)
    y2 = torch.maximum((cy + h / 2).clamp(0.0, float(img_h)), y1 + 1.3)
    boxes = torch.stack([x1, y1, x2, y2], dim=1).float()
    # knock the coordinates off any integer/half-integer lattice
    boxes = boxes + torch.rand(num_gt, 4, generator=gen) * 0.37 + 0.011
    labels = torch.randint(0, 80, (num_gt, ), generator=gen)
    return boxes.contiguous(), labels


def synthetic_batch(num_imgs=2,
                    img_shape=(800, 1333),
                    pad_shape=(800, 1344),
                    num_gt=7,
                    seed=1234,
                    device='cpu'):
    """One LD training batch in the mmdet batch contract
    (reference: mmdet/models/detectors/kd_one_stage.py:46-65).

    ``num_gt`` may be an int or a per-image list.
    """
    gen = _gen(seed)
    h, w = img_shape
    hp, wp = pad_shape
    img = torch.zeros(num_imgs, 3, hp, wp)
    img[:, :, :h, :w] = _normal((num_imgs, 3, h, w), gen)
    if isinstance(num_gt, int):
        num_gt = [num_gt] * num_imgs
    gt_bboxes, gt_labels = [], []
    for g in num_gt:
        b, l = synthetic_boxes(g, h, w, gen)
        gt_bboxes.append(b.to(device))
        gt_labels.append(l.to(device))
    img_metas = [
        dict(
            img_shape=(h, w, 3),
            pad_shape=(hp, wp, 3),
            ori_shape=(h, w, 3),
            scale_factor=1.0,
            flip=False) for _ in range(num_imgs)
    ]
    return dict(
        img=img.to(device),
        img_metas=img_metas,
        gt_bboxes=gt_bboxes,
        gt_labels=gt_labels)


def _key_seed(key, seed):
    return (zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF


def seeded_state_dict(reference_sd, seed=0, reg_std=0.05, cls_std=0.02):
    """Fill every entry of ``reference_sd`` (name -> tensor, used for shape and
    dtype only) with deterministic values keyed on the entry's *name*.

    The value distributions keep activations of a randomly initialised
    ResNet-FPN-GFL stack well conditioned (no overflow, non-degenerate
    softmaxes) so fp32 parity thresholds stay meaningful.
    """
    out = OrderedDict()
    for key, ref in reference_sd.items():
        g = _gen(_key_seed(key, seed))
        shape = tuple(ref.shape)
        leaf = key.rsplit('.', 1)[-1]
        parent = key.rsplit('.', 2)[-2] if key.count('.') >= 1 else ''
        if leaf == 'num_batches_tracked':
            v = torch.zeros(shape, dtype=ref.dtype)
        elif leaf == 'running_mean':
            v = torch.rand(shape, generator=g) * 0.2 - 0.1
        elif leaf == 'running_var':
            v = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif leaf == 'project':  # Integral buffer, gfl_head.py:29-30
            v = torch.linspace(0, shape[0] - 1, shape[0])
        elif leaf == 'scale':  # mmcv Scale
            v = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif len(shape) == 4:  # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            if parent == 'gfl_cls':
                std = cls_std
            elif parent in ('gfl_reg', 'reg_conf'):
                std = reg_std
            elif 'lateral_convs' in key:
                std = 0.5 * math.sqrt(1.0 / fan_in)
            elif 'fpn_convs' in key:
                std = math.sqrt(1.0 / fan_in)
            else:
                std = math.sqrt(2.0 / fan_in)
            v = _normal(shape, g) * std
        elif leaf == 'weight':  # norm affine
            block = key.rsplit('.', 2)[0]
            last = 'bn3' if block + '.bn3.weight' in reference_sd else 'bn2'
            is_last_bn = ('.layer' in key and parent == last) or \
                key.endswith('downsample.1.weight')
            if is_last_bn:
                v = torch.rand(shape, generator=g) * 0.2 + 0.2
            else:
                v = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif leaf == 'bias':
            if parent == 'gfl_cls':
                v = torch.rand(shape, generator=g) * 0.5 - 3.5
            else:
                v = torch.rand(shape, generator=g) * 0.2 - 0.1
        else:
            v = _normal(shape, g) * 0.1
        out[key] = v.to(ref.dtype).reshape(shape)
    return out


def level_shapes(pad_shape, strides=(8, 16, 32, 64, 128)):
    """Feature-map sizes of FPN P3..P7 for a padded input. P3..P5 follow the
    ResNet stride-2 convs (ceil for even pads), P6/P7 the 3x3 s2 pad-1 extra
    convs (reference: mmdet/models/necks/fpn.py:129-160)."""
    hp, wp = pad_shape

    def down(x):  # 3x3 stride 2 pad 1  ==  ceil(x / 2)
        return (x + 1) // 2

    h, w = hp, wp
    sizes = []
    for i in range(7):
        h, w = down(h), down(w)
        if i >= 2:
            sizes.append((h, w))
    assert len(sizes) == len(strides)
    return sizes


def synthetic_head_inputs(num_imgs,
                          featmap_sizes,
                          seed=0,
                          num_classes=80,
                          reg_max=16,
                          feat_channels=256,
                          device='cpu',
                          num_anchors=1):
    """Random student/teacher head outputs and FPN features for loss-block
    parity tests: logits ~ 3*randn (reg), cls ~ 1.2*randn - 4, features ~ randn.
    ``num_anchors`` > 1: anchor-major channel blocks (RetinaGFLHead).

    Returns dict of lists (one tensor per level, NCHW).
    """
    gen = _gen(seed)
    out = dict(cls=[], reg=[], t_cls=[], t_reg=[], x=[], t_x=[])
    na = num_anchors
    for (h, w) in featmap_sizes:
        out['cls'].append(
            _normal((num_imgs, na * num_classes, h, w), gen) * 1.2 - 4.0)
        out['reg'].append(
            _normal((num_imgs, na * 4 * (reg_max + 1), h, w), gen) * 3.0)
        out['t_cls'].append(
            _normal((num_imgs, na * num_classes, h, w), gen) * 1.2 - 4.0)
        out['t_reg'].append(
            _normal((num_imgs, na * 4 * (reg_max + 1), h, w), gen) * 3.0)
        out['x'].append(
            _normal((num_imgs, feat_channels, h, w), gen))
        out['t_x'].append(
            _normal((num_imgs, feat_channels, h, w), gen))
    return {k: [t.to(device) for t in v] for k, v in out.items()}


# ---------------------------------------------------------------------------
# inference cases (GFLHead.get_bboxes parity, tests/golden/infer.npz)
# ---------------------------------------------------------------------------
# name, pad, img_shapes, scale_factors, seed, nms_pre, cls_scale, cls_shift,
# store_all.  The class logits of synthetic_head_inputs are re-scaled to
# cls * cls_scale + cls_shift so the number of (anchor, class) pairs above
# score_thr = 0.05 lands on either side of batched_nms' split_thr = 10000.
INFER_CASES = [
    ('small', (128, 160), [(128, 160, 3), (120, 150, 3)],
     [[1.0, 1.0, 1.0, 1.0], [1.25, 1.25, 1.25, 1.25]], 51, 1000, 1.25, -1.0,
     True),
    ('small_topk', (128, 160), [(128, 160, 3), (100, 140, 3)],
     [[0.5, 0.5, 0.5, 0.5], [2.0, 2.0, 2.0, 2.0]], 52, 50, 1.25, -1.0, True),
    ('c2', (800, 1344), [(800, 1333, 3), (750, 1344, 3)],
     [[1.6675, 1.6675, 1.6675, 1.6675], [1.0, 1.0, 1.0, 1.0]], 53, 1000, 1.25,
     -1.0, False),
    ('c2_dense', (800, 1344), [(800, 1333, 3), (800, 1344, 3)],
     [[1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0]], 54, 1000, 1.0, 0.0, False),
]


def infer_inputs_clustered(pad, num_imgs, seed, device='cpu'):
    """Head outputs whose neighbouring anchors predict nearly the same box (one
    sharp distribution per side shared by all positions of a level plus small
    noise), so that score voting has clusters to average: the case the
    'voting_cluster_diounms' branch exists for.  IEEE-exact ops only."""
    sizes = level_shapes(pad)
    gen = _gen(seed)
    bins = torch.arange(17, dtype=torch.float32)
    cls, reg = [], []
    for li, (h, w) in enumerate(sizes):
        c = _normal((num_imgs, 80, h, w), gen) * 1.25 - 6.0
        c[:, 3] += 5.0   # two classes fire on every anchor: dense clusters of
        c[:, 17] += 5.0  # near-identical same-class boxes
        cls.append(c)
        centre = torch.tensor([5.0, 4.0, 6.0, 5.0]) + float(li % 2)
        peak = -((bins[None, :] - centre[:, None]) ** 2) * 2.0  # (4, 17)
        base = peak.reshape(1, 68, 1, 1).expand(num_imgs, 68, h, w)
        reg.append((base + _normal((num_imgs, 68, h, w), gen) * 0.3)
                   .contiguous())
    return [c.to(device) for c in cls], [r.to(device) for r in reg]


def synthetic_centerness(num_imgs, featmap_sizes, seed=0, device='cpu'):
    """Centerness logits ~ 1.5 * randn for the ATSS heads (a separate stream:
    the draw order of synthetic_head_inputs is part of the older goldens)."""
    gen = _gen(seed + 7919)
    return [(_normal((num_imgs, 1, h, w), gen) * 1.5).to(device)
            for (h, w) in featmap_sizes]


def grad_probe(n, seed):
    """Deterministic pseudo-random direction in [-0.5, 0.5)^n (integer hash,
    exact on every platform): gradient fingerprints dot(grad, probe) that --
    unlike a norm -- see sign flips, permutations and transposed layouts."""
    import numpy as np
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(seed * 40503 + 12345)) & \
        np.uint64(0xFFFFFFFF)
    h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(13))
    return h.astype(np.float64) / 4294967296.0 - 0.5


def grad_sample_idx(n, k=256, seed=7):
    """k deterministic flat indices into a gradient of n elements (all of them
    when n <= k): integer hash, exact on every platform, so the fixture and the
    test address the same elements without storing the indices."""
    import numpy as np
    if n <= k:
        return np.arange(n, dtype=np.int64)
    i = np.arange(k, dtype=np.uint64)
    h = (i * np.uint64(2246822519) + np.uint64(seed * 3266489917 % (1 << 32))) & \
        np.uint64(0xFFFFFFFF)
    h = (h ^ (h >> np.uint64(16))) * np.uint64(2654435761) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(13))
    return (h % np.uint64(n)).astype(np.int64)


VOTING_CASES = [
    # name, pad, img_shapes, scale factors, seed, nms_pre, clustered
    ('v_small', (128, 160), [(128, 160, 3), (120, 150, 3)],
     [[1.0, 1.0, 1.0, 1.0], [1.25, 1.25, 1.25, 1.25]], 61, 1000, False),
    ('v_clustered', (128, 160), [(128, 160, 3), (128, 150, 3)],
     [[1.0, 1.0, 1.0, 1.0], [2.0, 2.0, 2.0, 2.0]], 62, 1000, True),
    ('v_clustered_topk', (256, 320), [(256, 320, 3), (250, 300, 3)],
     [[1.0, 1.0, 1.0, 1.0], [0.5, 0.5, 0.5, 0.5]], 63, 60, True),
]


def voting_inputs(case, device='cpu'):
    import numpy as np
    name, pad, img_shapes, sfs, seed, nms_pre, clustered = case
    if clustered:
        cls, reg = infer_inputs_clustered(pad, len(img_shapes), seed, device)
    else:
        hi = synthetic_head_inputs(len(img_shapes), level_shapes(pad),
                                   seed=seed)
        cls = [(c * 1.25 - 1.0).to(device) for c in hi['cls']]
        reg = [r.to(device) for r in hi['reg']]
    metas = [dict(img_shape=s_, pad_shape=tuple(pad) + (3, ),
                  scale_factor=np.array(f, dtype=np.float32))
             for s_, f in zip(img_shapes, sfs)]
    return cls, reg, metas


def infer_inputs(case, device='cpu'):
    """(cls_scores, bbox_preds, img_metas) of an INFER_CASES row."""
    import numpy as np
    name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
    sizes = level_shapes(pad)
    hi = synthetic_head_inputs(len(img_shapes), sizes, seed=seed)
    cls = [(c * cs + sh).to(device) for c in hi['cls']]
    reg = [r.to(device) for r in hi['reg']]
    metas = [dict(img_shape=s_, pad_shape=tuple(pad) + (3, ),
                  scale_factor=np.array(f, dtype=np.float32))
             for s_, f in zip(img_shapes, sfs)]
    return cls, reg, metas


# ---------------------------------------------------------------------------
# GFocalHead.get_bboxes (GFLv2: the maps already hold probabilities, 81
# channels because use_sigmoid=False -> cls_out_channels = num_classes + 1)
# ---------------------------------------------------------------------------
INFER_V2_CASES = ['small', 'small_topk', 'c2']


def infer_inputs_prob(case, device='cpu'):
    """(cls_scores as probabilities (N, 81, H, W), bbox_preds, img_metas) of an
    INFER_CASES row: sigmoid of 81-channel synthetic logits (the sigmoid is
    part of the INPUT here, both paths receive the same floats)."""
    import numpy as np
    name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
    sizes = level_shapes(pad)
    hi = synthetic_head_inputs(len(img_shapes), sizes, seed=seed + 1000,
                               num_classes=81)
    cls = [torch.sigmoid(c * cs + sh).to(device) for c in hi['cls']]
    reg = [r.to(device) for r in hi['reg']]
    metas = [dict(img_shape=s_, pad_shape=tuple(pad) + (3, ),
                  scale_factor=np.array(f, dtype=np.float32))
             for s_, f in zip(img_shapes, sfs)]
    return cls, reg, metas


def infer_inputs_retina(case, device='cpu'):
    """(cls_scores (N, 9 * 80, H, W), bbox_preds (N, 9 * 68, H, W), img_metas)
    of an INFER_CASES row for the 9-anchor RetinaGFL head."""
    import numpy as np
    name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
    sizes = level_shapes(pad)
    hi = synthetic_head_inputs(len(img_shapes), sizes, seed=seed + 2000,
                               num_anchors=9)
    cls = [(c * cs + sh).to(device) for c in hi['cls']]
    reg = [r.to(device) for r in hi['reg']]
    metas = [dict(img_shape=s_, pad_shape=tuple(pad) + (3, ),
                  scale_factor=np.array(f, dtype=np.float32))
             for s_, f in zip(img_shapes, sfs)]
    return cls, reg, metas
