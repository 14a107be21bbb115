"""mmdet.core pieces on the LD path with their reference names and call
signatures: AnchorGenerator, BboxOverlaps2D,
ATSSAssigner (+ get_vlr_region), PseudoSampler, AssignResult, SamplingResult,
DeltaXYWHBBoxCoder (constructed by AnchorHead's default, never called here),
reduce_mean, distance2bbox / bbox2distance.  Arithmetic goes through libldhip.so.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import lib as L
from . import lossblock as LB
from .registry import (ANCHOR_GENERATORS, BBOX_ASSIGNERS, BBOX_CODERS,
                       BBOX_SAMPLERS, IOU_CALCULATORS, build_iou_calculator)

INF = 100000000


# --------------------------------------------------------------- utilities --
# The reference's host-side glue around target assignment -- multi_apply,
# unmap, images_to_levels, anchor_inside_flags (core/utils/misc.py:10-42,
# core/anchor/utils.py:4-46) -- has no counterpart here: ld_atss_targets writes
# the dense, level-major target tensors of the whole batch directly.
def reduce_mean(tensor):
    """mmdet/core/utils/dist_utils.py:63-69."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


def bbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py:99-116: (n, 5) detections + (n,) labels -> list
    of per-class (k_c, 5) float32 numpy arrays (the format evaluation and
    ``show_result`` consume)."""
    import numpy as np
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


_LTRB_SIGN = (-1.0, -1.0, 1.0, 1.0)


def distance2bbox(points, distance, max_shape=None):
    """(x, y) points + (l, t, r, b) distances -> xyxy boxes, optionally clipped
    to ``max_shape = (h, w)`` (semantics of core/bbox/transforms.py:119-156)."""
    boxes = points[:, (0, 1, 0, 1)] + distance * distance.new_tensor(_LTRB_SIGN)
    if max_shape is not None:
        hi = boxes.new_tensor((max_shape[1], max_shape[0]) * 2)
        boxes = torch.minimum(boxes.clamp(min=0), hi)
    return boxes


def bbox2distance(points, bbox, max_dis=None, eps=0.1):
    """xyxy boxes -> (l, t, r, b) distances from ``points``, clamped to
    [0, max_dis - eps] (semantics of core/bbox/transforms.py:159-180)."""
    dist_ = (bbox - points[:, (0, 1, 0, 1)]) * bbox.new_tensor(_LTRB_SIGN)
    if max_dis is not None:
        dist_ = dist_.clamp(min=0, max=max_dis - eps)
    return dist_


# ---------------------------------------------------------- IoU calculator --
_MODES = {'iou': 0, 'iof': 1, 'giou': 2, 'diou': 3}


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """mmdet/core/bbox/iou_calculators/iou2d_calculator.py:43-188 incl. the
    fork's IoF-based 'diou' mode; one HIP launch."""
    assert mode in _MODES, f'Unsupported mode {mode}'
    assert bboxes1.size(-1) == 4 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 4 or bboxes2.size(0) == 0
    if bboxes1.dim() != 2 or bboxes2.dim() != 2:
        raise NotImplementedError('batched bbox_overlaps')
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    L.require_device(bboxes1, torch.float32, 'bboxes1')
    L.require_device(bboxes2, torch.float32, 'bboxes2')
    shape = (rows, ) if is_aligned else (rows, cols)
    out = bboxes1.new_empty(shape)
    if rows * cols == 0:
        return out
    lib = L.get_lib()
    L.check(lib.ld_bbox_overlaps(L.ptr(bboxes1.contiguous()),
                                 L.ptr(bboxes2.contiguous()), rows, cols,
                                 _MODES[mode], 1 if is_aligned else 0, eps,
                                 L.ptr(out), L.stream_ptr(out.device)),
            'ld_bbox_overlaps')
    return out


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    """iou2d_calculator.py:10-35."""

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5]
        assert bboxes2.size(-1) in [0, 4, 5]
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        return bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + '()'


# --------------------------------------------------------- anchor generator --
@ANCHOR_GENERATORS.register_module()
class AnchorGenerator:
    """mmdet/core/anchor/anchor_generator.py:9-346, center_offset = 0.

    Two regimes:
      * one square anchor per cell (ratios=[1.0], scales_per_octave=1: every
        GFL / ATSS / LD config) -- the anchors are implicit in the kernels
        (side = octave_base_scale * stride, centred on the cell origin);
      * ratios x scales anchors per cell (RetinaGFLHead / LDRetinaHead) -- an
        explicit, cached (A * B, 4) list in the reference's order (cell, then
        base anchor; ratio-major base anchors), built with the reference's fp32
        expressions so MaxIoU thresholds compare the same bits."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None,
                 scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0 or centers is not None:
            raise NotImplementedError('center_offset / centers')
        if not scale_major:
            raise NotImplementedError('scale_major=False')
        if (octave_base_scale is not None and scales_per_octave is not None):
            if scales is not None:
                raise ValueError('scales and octave_base_scale with '
                                 'scales_per_octave cannot be set at the same '
                                 'time')
            scales = [2 ** (i / scales_per_octave) * octave_base_scale
                      for i in range(scales_per_octave)]
        elif scales is None:
            raise ValueError('Either scales or octave_base_scale with '
                             'scales_per_octave should be set')
        self.strides = [(s, s) if isinstance(s, int) else tuple(s)
                        for s in strides]
        for s in self.strides:
            if s[0] != s[1]:
                raise NotImplementedError('h stride != w stride')
        self.base_sizes = [min(s) for s in self.strides] \
            if base_sizes is None else list(base_sizes)
        if self.base_sizes != [min(s) for s in self.strides]:
            raise NotImplementedError('base_sizes != strides')
        self.ratios = torch.Tensor(list(ratios))
        self.scales = torch.Tensor(list(scales))
        self.single_square = (list(ratios) == [1.0] and len(scales) == 1 and
                              float(scales[0]) == int(scales[0]))
        self.anchor_scale = int(scales[0]) if self.single_square else None
        self.octave_base_scale = octave_base_scale
        self.scales_per_octave = scales_per_octave
        self.scale_major, self.centers = scale_major, centers
        self.center_offset = center_offset
        self._grid_cache = {}

    @property
    def num_base_anchors(self):
        return [self.ratios.numel() * self.scales.numel()
                for _ in self.strides]

    @property
    def num_levels(self):
        return len(self.strides)

    @property
    def base_anchors(self):
        """CPU tensors, as in the reference (gen_base_anchors,
        anchor_generator.py:142-185)."""
        out = []
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        for size in self.base_sizes:
            ws = (size * w_ratios[:, None] * self.scales[None, :]).view(-1)
            hs = (size * h_ratios[:, None] * self.scales[None, :]).view(-1)
            out.append(torch.stack([0. - 0.5 * ws, 0. - 0.5 * hs,
                                    0. + 0.5 * ws, 0. + 0.5 * hs], dim=-1))
        return out

    def grid_anchors(self, featmap_sizes, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        sizes = [tuple(int(v) for v in s) for s in featmap_sizes]
        if self.single_square:
            flat = LB.grid_anchors(sizes, [s[0] for s in self.strides],
                                   torch.device(device), self.anchor_scale)
            out, off = [], 0
            for h, w in sizes:
                out.append(flat[off:off + h * w])
                off += h * w
            return out
        key = (tuple(sizes), str(device))
        hit = self._grid_cache.get(key)
        if hit is None:
            # constants of the geometry, built once per pyramid shape
            hit = []
            for (h, w), s, base in zip(sizes, self.strides,
                                       self.base_anchors):
                sx = torch.arange(0, w, device=device) * s[0]
                sy = torch.arange(0, h, device=device) * s[1]
                xx, yy = sx.repeat(h), sy.view(-1, 1).repeat(1, w).view(-1)
                base = base.to(device)
                shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
                hit.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
            if len(self._grid_cache) > 16:
                self._grid_cache.clear()
            self._grid_cache[key] = hit
        return hit

    def grid_anchors_flat(self, featmap_sizes, device='cuda'):
        """All levels of :meth:`grid_anchors` as one cached (A * B, 4) tensor
        (what the batched target kernels take)."""
        sizes = [tuple(int(v) for v in s) for s in featmap_sizes]
        key = ('flat', tuple(sizes), str(device))
        hit = self._grid_cache.get(key)
        if hit is None:
            hit = torch.cat(list(self.grid_anchors(sizes, device))).contiguous()
            self._grid_cache[key] = hit
        return hit

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        """anchor_generator.py:272-328."""
        assert self.num_levels == len(featmap_sizes)
        flags = []
        for (fh, fw), s, nb in zip(featmap_sizes, self.strides,
                                   self.num_base_anchors):
            h, w = pad_shape[:2]
            vh = min(int(math.ceil(h / s[1])), int(fh))
            vw = min(int(math.ceil(w / s[0])), int(fw))
            f = torch.zeros((int(fh), int(fw)), dtype=torch.bool,
                            device=device)
            f[:vh, :vw] = True
            f = f.reshape(-1)
            if nb > 1:
                f = f[:, None].expand(f.numel(), nb).reshape(-1)
            flags.append(f)
        return flags

    def __repr__(self):
        return (f'{self.__class__.__name__}(strides={self.strides}, '
                f'ratios={self.ratios.tolist()}, '
                f'scales={self.scales.tolist()})')


# --------------------------------------------------- assign / sample results --
class AssignResult:
    """mmdet/core/bbox/assigners/assign_result.py."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels
        self._extra_properties = {}

    @property
    def num_preds(self):
        return len(self.gt_inds)


class SamplingResult:
    """Positive / negative index sets of one image with the boxes they select
    (attribute names of samplers/sampling_result.py:25-49, which LD code reads:
    pos_inds, neg_inds, pos_bboxes, neg_bboxes, pos_is_gt, num_gts,
    pos_assigned_gt_inds (0-based), pos_gt_bboxes, pos_gt_labels)."""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result,
                 gt_flags):
        gt = gt_bboxes.reshape(-1, 4)
        matched = assign_result.gt_inds[pos_inds] - 1  # 1-based -> 0-based
        if gt.shape[0] == 0 and matched.numel() != 0:
            raise AssertionError('positives without any ground-truth box')
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt.shape[0]
        self.pos_assigned_gt_inds = matched
        self.pos_gt_bboxes = gt[matched] if gt.shape[0] else gt
        lab = assign_result.labels
        self.pos_gt_labels = None if lab is None else lab[pos_inds]

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    """No sampling: every assigned anchor is positive, every unassigned one
    negative, both in ascending index order (what the reference obtains with
    nonzero().unique(), samplers/pseudo_sampler.py:24-41)."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        assigned = assign_result.gt_inds
        pos_inds = (assigned > 0).nonzero(as_tuple=True)[0]
        neg_inds = (assigned == 0).nonzero(as_tuple=True)[0]
        no_gt_proposals = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes,
                              assign_result, no_gt_proposals)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    """Constructed by AnchorHead's default bbox_coder (anchor_head.py:45-49)
    but never called on the GFL/LD path: registrable stub."""

    def __init__(self, target_means=(0., 0., 0., 0.),
                 target_stds=(1., 1., 1., 1.), clip_border=True):
        self.means, self.stds = target_means, target_stds
        self.clip_border = clip_border

    def encode(self, bboxes, gt_bboxes):
        raise NotImplementedError('DeltaXYWHBBoxCoder is not on the LD path')

    decode = encode


# ---------------------------------------------------------- MaxIoU assigner --
@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    """mmdet/core/bbox/assigners/max_iou_assigner.py:9-212 for the RetinaGFL /
    LDRetina configs: float thresholds, match_low_quality with
    gt_max_assign_all, no ignore regions.  ``assign`` runs the batched HIP
    target kernel (ld_retina_targets) on one image's explicit box list."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0,
                 gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True,
                 gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        if isinstance(neg_iou_thr, (tuple, list)):
            raise NotImplementedError('neg_iou_thr as an interval')
        if not (gt_max_assign_all and match_low_quality):
            raise NotImplementedError('gt_max_assign_all / match_low_quality '
                                      '= False')
        self.pos_iou_thr, self.neg_iou_thr = pos_iou_thr, neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr
        self.match_low_quality = match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None,
               gt_labels=None):
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and \
                gt_bboxes_ignore.numel() > 0:
            raise NotImplementedError('ignore regions (ignore_iof_thr > 0)')
        A = bboxes.size(0)
        t = LB.retina_targets(
            [(A, 1)], [1], [dict(pad_shape=(A, 1))], [gt_bboxes],
            [gt_labels if gt_labels is not None else
             gt_bboxes.new_zeros(gt_bboxes.size(0), dtype=torch.long)],
            [bboxes[:, :4].contiguous()], 1, self, 80, bboxes.device,
            want_gt_inds=True)
        gt_inds = t['gt_inds'][0]
        labels = None
        if gt_labels is not None:
            labels = torch.where(gt_inds > 0, t['labels'][0],
                                 torch.full_like(gt_inds, -1))
        return AssignResult(gt_bboxes.size(0), gt_inds, None, labels=labels)


# ------------------------------------------------------------ ATSS assigner --
@BBOX_ASSIGNERS.register_module()
class ATSSAssigner:
    """mmdet/core/bbox/assigners/atss_assigner.py:9-298, reference call
    signatures, on the batched HIP target kernels (explicit-anchor form)."""

    def __init__(self, topk, iou_calculator=dict(type='BboxOverlaps2D'),
                 ignore_iof_thr=-1):
        self.topk = topk
        self.iou_calculator = build_iou_calculator(iou_calculator)
        self.ignore_iof_thr = ignore_iof_thr

    def _run(self, bboxes, num_level_bboxes, gt_bboxes, gt_labels):
        if self.ignore_iof_thr > 0:
            raise NotImplementedError('ignore regions (ignore_iof_thr > 0)')
        lib = L.get_lib()
        bboxes = L.require_device(bboxes[:, :4].contiguous(), torch.float32,
                                  'bboxes')
        dev = bboxes.device
        A, G = bboxes.shape[0], gt_bboxes.shape[0]
        assert sum(num_level_bboxes) == A
        geom = L.make_geom([(int(n), 1) for n in num_level_bboxes],
                           [1] * len(num_level_bboxes), 1)
        hp = LB.make_hp(topk=self.topk)
        gtb = gt_bboxes.reshape(1, G, 4).contiguous() if G else \
            torch.zeros((1, 1, 4), device=dev)
        gtl = (gt_labels if gt_labels is not None else
               torch.zeros(G, dtype=torch.int64, device=dev))
        gtl = gtl.reshape(1, G).contiguous() if G else torch.zeros(
            (1, 1), dtype=torch.int64, device=dev)
        ng = LB._small_int_tensor((G, ), dev)
        vhw = LB._small_int_tensor(
            tuple((int(n), 1) for n in num_level_bboxes), dev)
        out = dict(
            labels=torch.empty((1, A), dtype=torch.int64, device=dev),
            lw=torch.empty((1, A), device=dev),
            bt=torch.empty((1, A, 4), device=dev),
            vlr=torch.empty((1, A), device=dev),
            im=torch.empty((1, A), device=dev),
            counts=torch.empty(1 + 2 * geom.num_levels + 1,
                               dtype=torch.int32, device=dev),
            gt_inds=torch.empty((1, A), dtype=torch.int64, device=dev),
            max_overlaps=torch.empty((1, A), device=dev))
        need = lib.ld_atss_targets_workspace_bytes(C.byref(geom), G)
        ws = LB.workspace(dev, need, 'targets')
        L.check(lib.ld_atss_targets_ex(
            C.byref(geom), C.byref(hp), L.ptr(bboxes), L.ptr(gtb), L.ptr(gtl),
            L.ptr(ng), G, L.ptr(vhw), L.ptr(out['labels']), L.ptr(out['lw']),
            L.ptr(out['bt']), L.ptr(out['vlr']), L.ptr(out['im']),
            L.ptr(out['counts']), L.ptr(out['gt_inds']),
            L.ptr(out['max_overlaps']), L.ptr(ws), ws.numel(),
            L.stream_ptr(dev)), 'ld_atss_targets_ex')
        return out

    def assign(self, bboxes, num_level_bboxes, gt_bboxes,
               gt_bboxes_ignore=None, gt_labels=None):
        num_gt, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        if num_gt == 0 or num_bboxes == 0:
            gt_inds = bboxes.new_full((num_bboxes, ), 0, dtype=torch.long)
            max_overlaps = bboxes.new_zeros((num_bboxes, ))
            labels = None if gt_labels is None else bboxes.new_full(
                (num_bboxes, ), -1, dtype=torch.long)
            return AssignResult(num_gt, gt_inds, max_overlaps, labels=labels)
        out = self._run(bboxes, num_level_bboxes, gt_bboxes, gt_labels)
        gt_inds = out['gt_inds'][0]
        labels = None
        if gt_labels is not None:
            # -1 for unassigned, as the reference (atss_assigner.py:170-177)
            labels = torch.where(gt_inds > 0, out['labels'][0],
                                 torch.full_like(gt_inds, -1))
        return AssignResult(num_gt, gt_inds, out['max_overlaps'][0],
                            labels=labels)

    def get_vlr_region(self, bboxes, num_level_bboxes, gt_bboxes,
                       gt_bboxes_ignore=None, gt_labels=None):
        """atss_assigner.py:183-298.  With no GT the reference returns an
        AssignResult by mistake (quirk Q2); this returns zeros."""
        if gt_bboxes.size(0) == 0 or bboxes.size(0) == 0:
            return bboxes.new_zeros((bboxes.size(0), ))
        return self._run(bboxes, num_level_bboxes, gt_bboxes,
                         gt_labels)['vlr'][0]
