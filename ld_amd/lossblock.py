"""Torch-facing wrappers of the fused target-assignment / loss-block kernels.

Host code here only marshals device pointers into the C ABI
(include/ld_hip.h); no arithmetic of the hot path happens in Python/ATen.
"""
import ctypes as C
import math

import torch

from . import lib as L

_WS = {}
# Buffers a larger one superseded.  They are kept alive for the life of the
# process: a hipGraph captured earlier has the old pointer baked into its
# launches (the wgrad side stream and the teacher stream are shared by every
# GraphedStep), and replaying it after the buffer was freed would write split-K
# slabs into memory the allocator has handed to someone else.  Sizes
# only grow, so the list stays short.
_WS_RETIRED = []
_WS_SCOPE = [None]


def workspace(device, nbytes, tag='ws'):
    """A cached scratch buffer (never shrinks), private to (device, current
    stream, tag): the teacher runs on its own stream concurrently with the
    student, so scratch must not be shared across streams."""
    # a launch list being recorded (detectors._teacher_replay) is replayed later on
    # whatever stream the teacher runs on: its scratch must belong to the LIST, not
    # to the stream it happened to be recorded on (round 5: lists recorded on the
    # capture stream and replayed on the teacher stream shared the student's
    # GroupNorm scratch -- an intermittent 1e-5 mismatch of every parameter)
    scope = _WS_SCOPE[0]
    key = (device, scope, tag) if scope is not None else \
        (device, L.stream_id(device), tag)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _WS_RETIRED.append(t)
        t = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8,
                        device=device)
        # a buffer born while a hipGraph is being captured lives in that graph's
        # private pool and dies with the graph: never keep it in this cache (a
        # later capture would bake a dangling pointer -- seen in round 3 as
        # corrupted weight gradients in the second graph test of a process)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _WS[key] = t
    return t


def level_views(x3, levels):
    """(N, C, P) -> list of (N, C, H_l, W_l) views (no copies, no autograd)."""
    N, c, _ = x3.shape
    outs, off = [], 0
    for h, w in levels:
        outs.append(x3[:, :, off:off + h * w].view(N, c, h, w))
        off += h * w
    return outs


def alloc_level_views(like):
    """Per-level (N, C, H_l, W_l) tensors that are views of ONE new (N, C, P)
    buffer (the loss block's gradient maps: SplitLevelsFn.backward then hands
    the buffer on without touching it)."""
    N, c = like[0].shape[:2]
    levels = tuple((int(t.shape[2]), int(t.shape[3])) for t in like)
    if any(tuple(t.shape[:2]) != (N, c) for t in like):
        return [torch.empty_like(t, memory_format=torch.contiguous_format)
                for t in like]
    buf = torch.empty((N, c, sum(h * w for h, w in levels)),
                      dtype=like[0].dtype, device=like[0].device)
    return level_views(buf, levels)


def make_hp(num_classes=80, reg_max=16, topk=9, feat_channels=256, lw_cls=1.0,
            qfl_beta=2.0, lw_bbox=2.0, giou_eps=1e-6, lw_dfl=0.25, lw_ld=0.25,
            T_ld=10.0, lw_ld_vlr=0.25, T_ld_vlr=10.0, lw_kd=10.0, T_kd=2.0,
            lw_im=2.0, cls_channels=0, flags=0, lw_ctr=0.0, focal_alpha=0.25):
    hp = L.LossHpT()
    hp.cls_channels, hp.flags = cls_channels, flags
    hp.lw_ctr, hp.focal_alpha = lw_ctr, focal_alpha
    hp.num_classes, hp.reg_max, hp.topk = num_classes, reg_max, topk
    hp.feat_channels = feat_channels
    hp.lw_cls, hp.qfl_beta, hp.lw_bbox, hp.giou_eps = (lw_cls, qfl_beta,
                                                       lw_bbox, giou_eps)
    hp.lw_dfl, hp.lw_ld, hp.T_ld = lw_dfl, lw_ld, T_ld
    hp.lw_ld_vlr, hp.T_ld_vlr, hp.lw_kd, hp.T_kd = (lw_ld_vlr, T_ld_vlr,
                                                     lw_kd, T_kd)
    hp.lw_im = lw_im
    return hp


def valid_hw_from_metas(featmap_sizes, strides, img_metas):
    """anchor_generator.py:293-300: valid_h = min(ceil(pad_h / stride), H)."""
    rows = []
    for meta in img_metas:
        ph, pw = meta['pad_shape'][:2]
        for (h, w), s in zip(featmap_sizes, strides):
            s = s[0] if isinstance(s, (tuple, list)) else s
            rows.append([min(int(math.ceil(ph / s)), h),
                         min(int(math.ceil(pw / s)), w)])
    return rows


_INT_CACHE = {}


def _small_int_tensor(values, device):
    """Host integers -> device int32 tensor WITHOUT a synchronising pageable
    H2D copy (which would stall the host until the whole forward has drained):
    values seen before are served from a cache, new ones go through pinned
    memory with a non-blocking copy."""
    key = (str(device), values)
    t = _INT_CACHE.get(key)
    if t is None:
        if len(_INT_CACHE) > 4096:
            _INT_CACHE.clear()
        host = torch.tensor(values, dtype=torch.int32).pin_memory()
        t = host.to(device, non_blocking=True)
        _INT_CACHE[key] = (t, host)  # keep the pinned source alive
        return t
    return t[0]


class PinnedRing:
    """A few pinned host staging buffers in rotation for small host -> device
    updates that are enqueued asynchronously: the host must not rewrite a pinned
    buffer before the copy that reads it has EXECUTED (with graph replays queued
    ahead of it that can be a whole step later -- round 3 found exactly this
    race: step i training with step i + 1's box counts).  Each slot carries the
    event of its last copy; a slot is rewritten only after that event."""

    def __init__(self, numel, dtype, depth=4):
        self.bufs = [torch.zeros(numel, dtype=dtype).pin_memory()
                     for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def stage(self, dst, values):
        """dst (device tensor) <- values (host tensor / sequence), async."""
        k = self.i
        self.i = (k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.bufs[k].copy_(torch.as_tensor(values, dtype=self.bufs[k].dtype)
                           .reshape(-1))
        dst.copy_(self.bufs[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


class StaticTargets:
    """Fixed-address ground-truth / valid-region buffers for a captured train
    step (train.GraphedStep).  A hipGraph replays launches with their pointer
    and by-value arguments frozen, while the reference's batches carry a
    different number of boxes every iteration
    (mmdet/models/detectors/kd_one_stage.py:52-65): so the boxes live in
    (N, max_gt, 4) / (N, max_gt) buffers padded to ``max_gt``, the per-image
    COUNT lives on the device (the target kernels already loop to a run-time
    count and index with the ``max_gt`` stride), and the per-level valid region
    of ``img_metas[i]['pad_shape']`` is a device buffer too.  ``load`` rewrites
    all of them from a new batch -- outside the graph, before a replay.
    Attached to ``img_metas[0]['ld_static_targets']``; the target wrappers below
    then use these buffers instead of building their own."""

    def __init__(self, num_imgs, max_gt, device):
        self.N, self.max_gt, self.device = int(num_imgs), int(max_gt), device
        if self.max_gt < 1:
            raise ValueError('max_gt must be >= 1')
        self.gtb = torch.zeros((self.N, self.max_gt, 4), dtype=torch.float32,
                               device=device)
        self.gtl = torch.zeros((self.N, self.max_gt), dtype=torch.int64,
                               device=device)
        self.ng = torch.zeros(self.N, dtype=torch.int32, device=device)
        self._ng_ring = PinnedRing(self.N, torch.int32)
        self.vhw = self._vhw_ring = None
        self.geometry = None   # (featmap_sizes, strides) the vhw buffer is for
        self.metas = None
        self.num_gt = [0] * self.N

    def load(self, img_metas, gt_bboxes, gt_labels):
        if len(img_metas) != self.N or len(gt_bboxes) != self.N:
            raise ValueError(f'StaticTargets holds {self.N} images')
        num_gt = [int(b.shape[0]) for b in gt_bboxes]
        if max(num_gt) > self.max_gt:
            raise ValueError(f'{max(num_gt)} GT boxes in one image, the '
                             f'captured step holds at most {self.max_gt}')
        for i, (b, l) in enumerate(zip(gt_bboxes, gt_labels)):
            if b.data_ptr() == self.gtb[i].data_ptr():
                continue  # the caller handed our own padded views back
            if num_gt[i]:
                self.gtb[i, :num_gt[i]].copy_(b, non_blocking=True)
                self.gtl[i, :num_gt[i]].copy_(l, non_blocking=True)
            self.num_gt[i] = num_gt[i]
        self._ng_ring.stage(self.ng, self.num_gt)
        self.metas = [dict((k, v) for k, v in m.items()
                           if k != 'ld_static_targets') for m in img_metas]
        if self.geometry is not None:
            self._fill_vhw()

    def views(self):
        """Per-image (max_gt, 4) / (max_gt,) views with the mmdet list layout."""
        return ([self.gtb[i] for i in range(self.N)],
                [self.gtl[i] for i in range(self.N)])

    def _fill_vhw(self):
        rows = valid_hw_from_metas(self.geometry[0], self.geometry[1],
                                   self.metas)
        flat = torch.tensor(rows, dtype=torch.int32).reshape(-1)
        if self.vhw is None or self.vhw.numel() != flat.numel():
            if torch.cuda.is_current_stream_capturing():
                raise L.LdError('StaticTargets: the valid-region buffer must '
                                'exist before capture (run a warm-up step)')
            self.vhw = torch.zeros(flat.numel(), dtype=torch.int32,
                                   device=self.device)
            self._vhw_ring = PinnedRing(flat.numel(), torch.int32)
        self._vhw_ring.stage(self.vhw, flat)

    def valid_hw(self, featmap_sizes, strides):
        key = (tuple(tuple(s) for s in featmap_sizes),
               tuple(s[0] if isinstance(s, (tuple, list)) else s
                     for s in strides))
        if self.geometry is None or self.geometry != key:
            if torch.cuda.is_current_stream_capturing():
                raise L.LdError('StaticTargets: pyramid geometry changed '
                                'inside a capture')
            self.geometry = key
            self._fill_vhw()
        return self.vhw


def _static_targets(img_metas):
    return img_metas[0].get('ld_static_targets') if img_metas else None


def atss_targets(featmap_sizes, strides, img_metas, gt_bboxes, gt_labels, hp,
                 device, anchor_scale=8):
    """Dense ATSS / VLR / IM targets for a batch (see ld_atss_targets in
    include/ld_hip.h).  Returns a dict of device tensors + the geometry."""
    lib = L.get_lib()
    N = len(img_metas)
    geom = L.make_geom(featmap_sizes, strides, N, anchor_scale)
    A, nl = geom.num_anchors, geom.num_levels
    st = _static_targets(img_metas)
    if st is not None:
        # captured step: padded boxes + device-side counts at fixed addresses
        gtb, gtl, ng, max_gt = st.gtb, st.gtl, st.ng, st.max_gt
        num_gt = list(st.num_gt)
        vhw = st.valid_hw(featmap_sizes, strides)
    else:
        num_gt = [int(b.shape[0]) for b in gt_bboxes]
        max_gt = max(num_gt) if num_gt else 0
        gtb = torch.zeros((N, max(max_gt, 1), 4), dtype=torch.float32,
                          device=device)
        gtl = torch.zeros((N, max(max_gt, 1)), dtype=torch.int64,
                          device=device)
        for i, (b, l) in enumerate(zip(gt_bboxes, gt_labels)):
            if num_gt[i]:
                L.require_device(b, torch.float32, 'gt_bboxes')
                gtb[i, :num_gt[i]] = b
                gtl[i, :num_gt[i]] = l
        ng = _small_int_tensor(tuple(num_gt), device)
        vhw = _small_int_tensor(
            tuple(tuple(r) for r in valid_hw_from_metas(featmap_sizes, strides,
                                                        img_metas)), device)
    out = dict(
        labels=torch.empty((N, A), dtype=torch.int64, device=device),
        label_weights=torch.empty((N, A), dtype=torch.float32, device=device),
        bbox_targets=torch.empty((N, A, 4), dtype=torch.float32,
                                 device=device),
        vlr=torch.empty((N, A), dtype=torch.float32, device=device),
        im=torch.empty((N, A), dtype=torch.float32, device=device),
        counts=torch.empty(N + 2 * nl + 1, dtype=torch.int32, device=device),
    )
    need = lib.ld_atss_targets_workspace_bytes(C.byref(geom), max_gt)
    ws = workspace(device, need, 'targets')
    rc = lib.ld_atss_targets(
        C.byref(geom), C.byref(hp), L.ptr(gtb), L.ptr(gtl), L.ptr(ng), max_gt,
        L.ptr(vhw), L.ptr(out['labels']), L.ptr(out['label_weights']),
        L.ptr(out['bbox_targets']), L.ptr(out['vlr']), L.ptr(out['im']),
        L.ptr(out['counts']), L.ptr(ws), ws.numel(), L.stream_ptr(device))
    L.check(rc, 'ld_atss_targets')
    out['geom'] = geom
    out['num_gt'] = num_gt
    return out


def fcos_targets(featmap_sizes, strides, gt_bboxes, gt_labels, num_classes,
                 regress_ranges, center_sampling, center_sample_radius,
                 device):
    """Dense FCOS point targets for a batch (ld_fcos_targets): the same dict as
    :func:`atss_targets` (bbox_targets = (l, t, r, b) distances, vlr = the
    head's "remain" points)."""
    lib = L.get_lib()
    N = len(gt_bboxes)
    geom = L.make_geom(featmap_sizes, strides, N, 8)
    A, nl = geom.num_anchors, geom.num_levels
    num_gt = [int(b.shape[0]) for b in gt_bboxes]
    max_gt = max(num_gt) if num_gt else 0
    gtb = torch.zeros((N, max(max_gt, 1), 4), dtype=torch.float32,
                      device=device)
    gtl = torch.zeros((N, max(max_gt, 1)), dtype=torch.int64, device=device)
    for i, (b, l) in enumerate(zip(gt_bboxes, gt_labels)):
        if num_gt[i]:
            L.require_device(b, torch.float32, 'gt_bboxes')
            gtb[i, :num_gt[i]] = b
            gtl[i, :num_gt[i]] = l
    ng = _small_int_tensor(tuple(num_gt), device)
    out = dict(
        labels=torch.empty((N, A), dtype=torch.int64, device=device),
        label_weights=torch.empty((N, A), dtype=torch.float32, device=device),
        bbox_targets=torch.empty((N, A, 4), dtype=torch.float32,
                                 device=device),
        vlr=torch.empty((N, A), dtype=torch.float32, device=device),
        im=torch.empty((N, A), dtype=torch.float32, device=device),
        counts=torch.empty(N + 2 * nl + 1, dtype=torch.int32, device=device),
    )
    flat = [float(min(v, 1e8)) for r in regress_ranges for v in r]
    assert len(flat) == 2 * nl
    rr = (C.c_float * len(flat))(*flat)
    L.check(lib.ld_fcos_targets(
        C.byref(geom), int(num_classes), rr, 1 if center_sampling else 0,
        float(center_sample_radius), L.ptr(gtb), L.ptr(gtl), L.ptr(ng), max_gt,
        L.ptr(out['labels']), L.ptr(out['label_weights']),
        L.ptr(out['bbox_targets']), L.ptr(out['vlr']), L.ptr(out['im']),
        L.ptr(out['counts']), L.stream_ptr(device)), 'ld_fcos_targets')
    out['geom'] = geom
    out['num_gt'] = num_gt
    return out


def retina_targets(featmap_sizes, strides, img_metas, gt_bboxes, gt_labels,
                   anchors, num_base, assigner, num_classes, device,
                   want_gt_inds=False):
    """MaxIoU + VLR targets of a ``num_base``-anchors-per-cell head for a batch
    (ld_retina_targets).  ``anchors``: the per-level explicit lists of
    AnchorGenerator.grid_anchors.  Arrays come back in the pseudo-image layout
    (N * num_base, A) the loss block sweeps (include/ld_hip.h LD_LOSS_RETINA);
    ``geom`` in the result is that pseudo-image geometry."""
    lib = L.get_lib()
    N, B = len(img_metas), int(num_base)
    geom = L.make_geom(featmap_sizes, strides, N, 8)
    A, nl = geom.num_anchors, geom.num_levels
    flat = anchors[0] if len(anchors) == 1 else torch.cat(list(anchors))
    flat = L.require_device(flat.contiguous(), torch.float32, 'anchors')
    if flat.shape != (A * B, 4):
        raise L.LdError(f'retina_targets: {tuple(flat.shape)} anchors for '
                        f'{A} cells x {B}')
    num_gt = [int(b.shape[0]) for b in gt_bboxes]
    max_gt = max(num_gt) if num_gt else 0
    gtb = torch.zeros((N, max(max_gt, 1), 4), dtype=torch.float32,
                      device=device)
    gtl = torch.zeros((N, max(max_gt, 1)), dtype=torch.int64, device=device)
    for i, (b, l) in enumerate(zip(gt_bboxes, gt_labels)):
        if num_gt[i]:
            L.require_device(b, torch.float32, 'gt_bboxes')
            gtb[i, :num_gt[i]] = b
            gtl[i, :num_gt[i]] = l
    ng = _small_int_tensor(tuple(num_gt), device)
    vhw = _small_int_tensor(
        tuple(tuple(r) for r in valid_hw_from_metas(featmap_sizes, strides,
                                                    img_metas)), device)
    NB = N * B
    out = dict(
        labels=torch.empty((NB, A), dtype=torch.int64, device=device),
        label_weights=torch.empty((NB, A), dtype=torch.float32, device=device),
        bbox_targets=torch.empty((NB, A, 4), dtype=torch.float32,
                                 device=device),
        vlr=torch.empty((NB, A), dtype=torch.float32, device=device),
        im=torch.empty((NB, A), dtype=torch.float32, device=device),
        counts=torch.empty(NB + 2 * nl + 1, dtype=torch.int32, device=device),
    )
    if want_gt_inds:
        out['gt_inds'] = torch.empty((NB, A), dtype=torch.int64,
                                     device=device)
    need = lib.ld_retina_targets_workspace_bytes(C.byref(geom), B, max_gt)
    ws = workspace(device, need, 'targets')
    L.check(lib.ld_retina_targets(
        C.byref(geom), B, L.ptr(flat), int(num_classes),
        float(assigner.pos_iou_thr), float(assigner.neg_iou_thr),
        float(assigner.min_pos_iou), 9, L.ptr(gtb), L.ptr(gtl), L.ptr(ng),
        max_gt, L.ptr(vhw), L.ptr(out['labels']), L.ptr(out['label_weights']),
        L.ptr(out['bbox_targets']), L.ptr(out['vlr']), L.ptr(out['im']),
        L.ptr(out['counts']), L.ptr(out.get('gt_inds')), L.ptr(ws),
        ws.numel(), L.stream_ptr(device)), 'ld_retina_targets')
    out['geom'] = L.make_geom(featmap_sizes, strides, NB, 8)
    out['num_gt'] = num_gt
    out['num_base'] = B
    return out


def grid_anchors(featmap_sizes, strides, device, anchor_scale=8):
    lib = L.get_lib()
    geom = L.make_geom(featmap_sizes, strides, 1, anchor_scale)
    out = torch.empty((geom.num_anchors, 4), dtype=torch.float32,
                      device=device)
    L.check(lib.ld_grid_anchors(C.byref(geom), L.ptr(out),
                                L.stream_ptr(device)), 'ld_grid_anchors')
    return out


def gi_region(hp, targets, cls, reg, t_cls, t_reg, topn=10, iou_thr=0.3):
    """The 'gibox' imitation region (ld_gi_region): replaces targets['im'] and
    the per-level IM counts in targets['counts'] by the (<= topn per level)
    General-Instance boxes' cells.  Returns the updated targets dict (a shallow
    copy: the assignment results themselves are shared)."""
    lib = L.get_lib()
    geom = targets['geom']
    device = cls[0].device
    im = torch.empty_like(targets['im'])
    counts = targets['counts'].clone()
    ws = workspace(device, lib.ld_gi_region_workspace_bytes(C.byref(geom)),
                   'gi_region')
    L.check(lib.ld_gi_region(
        C.byref(geom), C.byref(hp), C.byref(L.make_maps(cls)),
        C.byref(L.make_maps(reg)), C.byref(L.make_maps(t_cls)),
        C.byref(L.make_maps(t_reg)), int(topn), float(iou_thr), L.ptr(im),
        L.ptr(counts), L.ptr(ws), ws.numel(), L.stream_ptr(device)),
        'ld_gi_region')
    out = dict(targets)
    out['im'], out['counts'] = im, counts
    return out


class _LossState:
    pass


def loss_block_forward(hp, targets, cls, reg, t_cls, t_reg, x, t_x,
                       reduce_norm=None, upstream=None, kd_s=None, kd_t=None,
                       ctr=None):
    """Run prepass -> (normaliser reduction) -> main -> finalise.

    cls/reg/x: student per-level NCHW tensors; t_*: teacher's.
    kd_s / kd_t (LDv2): student / teacher maps of the KD term when it does not
    run on ``cls`` itself (raw cls_feat, ld_gflv2.py:243); ``cls`` then holds
    probabilities and ``t_cls`` is unused.
    ctr (LDATSSHead, hp.flags & LD_LOSS_ATSS): the student's centerness maps;
    their gradient comes back as grads['ctr'] and the centerness loss in row 6.
    reduce_norm: optional callable(norm_tensor[2]) doing the cross-rank MEAN
    in place (core/utils/dist_utils.py:63-69) -- device side, no host sync.
    Returns (losses (8, L) tensor, grads dict of lists, norm tensor).
    """
    lib = L.get_lib()
    geom = targets['geom']
    device = cls[0].device
    N, A = geom.num_imgs, geom.num_anchors
    m_cls, m_reg = L.make_maps(cls), L.make_maps(reg)
    m_tcls, m_treg = L.make_maps(t_cls), L.make_maps(t_reg)
    m_x, m_tx = L.make_maps(x), L.make_maps(t_x)
    # gradient maps = the level views of ONE (N, C, P) buffer each: what
    # layers.SplitLevelsFn hands back to the level-concatenated head tensors
    # without a copy
    g_cls = alloc_level_views(cls)
    g_reg = alloc_level_views(reg)
    g_x = alloc_level_views(x)
    mg_cls, mg_reg, mg_x = L.make_maps(g_cls), L.make_maps(g_reg), \
        L.make_maps(g_x)
    split = kd_s is not None
    g_kd = m_kds = m_kdt = mg_kd = None
    if split:
        g_kd = alloc_level_views(kd_s)
        m_kds, m_kdt, mg_kd = (L.make_maps(kd_s), L.make_maps(kd_t),
                               L.make_maps(g_kd))
    wt = torch.empty((N, A), dtype=torch.float32, device=device)
    score = torch.empty((N, A), dtype=torch.float32, device=device)
    norm = torch.zeros(4, dtype=torch.float32, device=device)
    losses = torch.empty((L.LD_NUM_LOSS_KEYS, geom.num_levels),
                         dtype=torch.float32, device=device)
    need = lib.ld_loss_workspace_bytes(C.byref(geom))
    ws = workspace(device, need, 'loss')
    st = L.stream_ptr(device)
    L.check(lib.ld_loss_prepass_ex(
        C.byref(geom), C.byref(hp), C.byref(m_cls), C.byref(m_reg),
        L.ptr(targets['labels']), L.ptr(targets['bbox_targets']),
        L.ptr(targets['vlr']), L.ptr(targets['counts']), L.ptr(wt),
        L.ptr(score), L.ptr(norm), L.ptr(ws), ws.numel(), st),
        'ld_loss_prepass')
    if reduce_norm is not None:
        reduce_norm(norm)
    up = None
    if upstream is not None:
        up = L.require_device(upstream.contiguous(), torch.float32,
                              'upstream')
    L.check(lib.ld_loss_main_parts(
        C.byref(geom), C.byref(hp), C.byref(m_cls), C.byref(m_reg),
        C.byref(m_tcls), C.byref(m_treg), C.byref(m_x), C.byref(m_tx),
        L.ptr(targets['labels']), L.ptr(targets['label_weights']),
        L.ptr(targets['bbox_targets']), L.ptr(targets['vlr']),
        L.ptr(targets['im']), L.ptr(targets['counts']), L.ptr(wt),
        L.ptr(score), L.ptr(norm), L.ptr(up), C.byref(mg_cls),
        C.byref(mg_reg), C.byref(mg_x),
        C.byref(m_kds) if split else None, C.byref(m_kdt) if split else None,
        C.byref(mg_kd) if split else None, L.ptr(ws), ws.numel(), 15, st),
        'ld_loss_main')
    g_ctr = None
    if hp.flags & L.LD_LOSS_ATSS:
        if ctr is None:
            raise L.LdError('LD_LOSS_ATSS needs the centerness maps')
        g_ctr = alloc_level_views(ctr)
        L.check(lib.ld_loss_centerness(
            C.byref(geom), C.byref(hp), C.byref(L.make_maps(ctr)),
            L.ptr(targets['labels']), L.ptr(score), L.ptr(norm), L.ptr(up),
            C.byref(L.make_maps(g_ctr)), L.ptr(ws), ws.numel(), st),
            'ld_loss_centerness')
    L.check(lib.ld_loss_finalize(
        C.byref(geom), C.byref(hp), L.ptr(targets['counts']), L.ptr(norm),
        L.ptr(ws), L.ptr(losses), st), 'ld_loss_finalize')
    return losses, dict(cls=g_cls, reg=g_reg, x=g_x, kd=g_kd, ctr=g_ctr), \
        norm, dict(weight_targets=wt, score=score)


class LDLossBlock(torch.autograd.Function):
    """losses(8, L) = f(student cls[L], reg[L], x[L]); teacher tensors and
    targets are constants.  The gradient is produced by the same fused launch
    as the forward (upstream = 1); if the incoming grad is not all-ones the
    block is re-run with the actual upstream coefficients."""

    @staticmethod
    def forward(ctx, hp, targets, teacher, reduce_norm, unit_upstream,
                *student):
        nl = targets['geom'].num_levels
        cls, reg, x = (student[:nl], student[nl:2 * nl],
                       student[2 * nl:3 * nl])
        # 4th student group: LDv2 = raw cls_feat, LDATSS = centerness maps
        extra = student[3 * nl:] or None
        atss = bool(hp.flags & L.LD_LOSS_ATSS)
        kd_s, ctr = (None, extra) if atss else (extra, None)
        t_cls, t_reg, t_x = teacher[:3]
        kd_t = teacher[3] if len(teacher) > 3 else None
        losses, grads, norm, aux = loss_block_forward(
            hp, targets, cls, reg, t_cls, t_reg, x, t_x, reduce_norm,
            kd_s=kd_s, kd_t=kd_t, ctr=ctr)
        ctx.unit_upstream = unit_upstream
        ctx.pack = (hp, targets, teacher, norm)
        ctx.nl = nl
        ctx.save_for_backward(*student)
        ctx.grads = grads
        ctx.mark_non_differentiable(norm)
        return losses, norm

    @staticmethod
    def backward(ctx, g_losses, _g_norm):
        grads = ctx.grads
        if not ctx.unit_upstream:
            hp, targets, teacher, norm = ctx.pack
            student = ctx.saved_tensors
            nl = ctx.nl
            cls, reg, x = (student[:nl], student[nl:2 * nl],
                           student[2 * nl:3 * nl])
            extra = student[3 * nl:] or None
            atss = bool(hp.flags & L.LD_LOSS_ATSS)
            kd_s, ctr = (None, extra) if atss else (extra, None)
            # the normalisers are constants of the graph (the reference takes
            # them through .item(), ld_head.py:340-341,363): reuse them
            _, grads, _, _ = _rerun_with_upstream(hp, targets, teacher, norm,
                                                  cls, reg, x, g_losses, kd_s,
                                                  ctr)
        out = tuple(grads['cls']) + tuple(grads['reg']) + tuple(grads['x'])
        if grads.get('kd') is not None:
            out = out + tuple(grads['kd'])
        if grads.get('ctr') is not None:
            out = out + tuple(grads['ctr'])
        return (None, None, None, None, None) + out


def _rerun_with_upstream(hp, targets, teacher, norm, cls, reg, x, upstream,
                         kd_s=None, ctr=None):
    t_cls, t_reg, t_x = teacher[:3]
    kd_t = teacher[3] if len(teacher) > 3 else None
    fixed = norm.clone()

    def _restore(n):
        n.copy_(fixed)

    return loss_block_forward(hp, targets, cls, reg, t_cls, t_reg, x, t_x,
                              reduce_norm=_restore, upstream=upstream,
                              kd_s=kd_s, kd_t=kd_t, ctr=ctr)


# ---------------------------------------------------------------------------
# the north-star kernel on its own
# ---------------------------------------------------------------------------
def kl_integral_dense(s_reg, t_reg, weight, T=10.0, scale=1.0, with_grad=True):
    """(68, rows) student/teacher logits, (rows,) weight ->
    integral (4, rows), loss_rows (4, rows) = weight * KL per side,
    grad (68, rows) | None."""
    lib = L.get_lib()
    for t in (s_reg, t_reg, weight):
        L.require_device(t, torch.float32)
    rows = s_reg.shape[1]
    assert s_reg.shape[0] == 68 and s_reg.is_contiguous()
    integral = torch.empty((4, rows), dtype=torch.float32,
                           device=s_reg.device)
    loss_rows = torch.empty((4, rows), dtype=torch.float32,
                            device=s_reg.device)
    grad = torch.empty_like(s_reg) if with_grad else None
    L.check(lib.ld_kl_integral_dense(
        L.ptr(s_reg), L.ptr(t_reg), L.ptr(weight), rows, T, scale,
        L.ptr(integral), L.ptr(loss_rows), L.ptr(grad),
        L.stream_ptr(s_reg.device)), 'ld_kl_integral_dense')
    return integral, loss_rows, grad


# ---------------------------------------------------------------------------
# inference post-processing
# ---------------------------------------------------------------------------
def get_bboxes(cls_scores, bbox_preds, strides, img_shapes, scale_factors=None,
               nms_pre=1000, score_thr=0.05, iou_thr=0.6, max_per_img=100,
               num_classes=None, reg_max=16, voting=False, prob=False,
               centernesses=None, points=False, num_base=1, with_nms=True):
    """GFLHead.get_bboxes on the device (ld_get_bboxes_ex; ``voting`` = the
    score-voting Cluster-DIoU-NMS variant, ``prob`` = the class maps hold
    probabilities: GFocalHead.get_bboxes; ``centernesses`` / ``points``: the
    ATSSGFLHead / FCOSGFLHead variants).  ``cls_scores`` /
    ``bbox_preds``: per-level NCHW maps; ``img_shapes``: per image (h, w[, c]);
    ``scale_factors``: per image 4 values (rescale=True) or None.
    -> list of (dets (k, 5), labels (k,)) device tensors, one pair per image."""
    lib = L.get_lib()
    dev = cls_scores[0].device
    N = cls_scores[0].shape[0]
    B_ = int(num_base)
    C_ = int(num_classes or cls_scores[0].shape[1] // B_)
    if cls_scores[0].shape[1] != B_ * C_ or \
            bbox_preds[0].shape[1] != B_ * 4 * (reg_max + 1):
        raise L.LdError('get_bboxes: channel counts do not match num_base x '
                        '(num_classes, 4 * (reg_max + 1))')
    sizes = [tuple(int(v) for v in c.shape[-2:]) for c in cls_scores]
    g = L.make_geom(sizes, strides, N)
    cm, rm = L.make_maps(cls_scores), L.make_maps(bbox_preds)
    hw = torch.tensor([[float(s[0]), float(s[1])] for s in img_shapes],
                      dtype=torch.float32).to(dev)
    sf = None
    if scale_factors is not None:
        sf = torch.tensor([[float(v) for v in f] for f in scale_factors],
                          dtype=torch.float32).to(dev)
    need = lib.ld_get_bboxes_ex_workspace_bytes(C.byref(g), C_, B_,
                                                int(nms_pre))
    if need == 0:
        raise L.LdError('ld_get_bboxes: bad geometry')
    ws = workspace(dev, need, 'infer')
    pflags = (L.LD_INFER_PROB if prob else 0) | \
        (L.LD_INFER_POINTS if points else 0)
    flags = pflags | (L.LD_INFER_VOTING if voting else 0)
    km = C.byref(L.make_maps(centernesses)) if centernesses is not None \
        else None
    if not with_nms:
        # get_bboxes(with_nms=False): per image (mlvl_bboxes (K, 4), mlvl_scores
        # (K, C + 1) with the reference's zero background column[, factors (K,)])
        if voting:
            raise L.LdError('with_nms=False has no nms type')
        K = lib.ld_get_bboxes_num_selected(C.byref(g), B_, int(nms_pre))
        boxes = torch.empty((N, K, 4), dtype=torch.float32, device=dev)
        scores = torch.zeros((N, K, C_ + 1), dtype=torch.float32, device=dev)
        raw = torch.empty((N, K, C_), dtype=torch.float32, device=dev)
        fac = torch.empty((N, K), dtype=torch.float32, device=dev) \
            if centernesses is not None else None
        L.check(lib.ld_get_bboxes_pre_nms(
            C.byref(g), C.byref(cm), C.byref(rm), km, C_, B_, int(reg_max),
            L.ptr(hw), L.ptr(sf), int(nms_pre), pflags, L.ptr(boxes),
            L.ptr(raw), L.ptr(fac), L.ptr(ws), ws.numel(),
            L.stream_ptr(dev)), 'ld_get_bboxes_pre_nms')
        scores[:, :, :C_] = raw
        if fac is not None:
            return [(boxes[n], scores[n], fac[n]) for n in range(N)]
        return [(boxes[n], scores[n]) for n in range(N)]
    dets = torch.empty((N, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((N, max_per_img), dtype=torch.int64, device=dev)
    counts = torch.empty((N, ), dtype=torch.int32, device=dev)
    L.check(lib.ld_get_bboxes_ex(
        C.byref(g), C.byref(cm), C.byref(rm), km, C_, B_, int(reg_max),
        L.ptr(hw),
        L.ptr(sf), int(nms_pre), float(score_thr), float(iou_thr),
        int(max_per_img), flags, L.ptr(dets), L.ptr(labels), L.ptr(counts),
        L.ptr(ws), ws.numel(), L.stream_ptr(dev)), 'ld_get_bboxes_ex')
    ks = counts.cpu().tolist()  # the one sync of the call
    return [(dets[n, :k], labels[n, :k]) for n, k in enumerate(ks)]
