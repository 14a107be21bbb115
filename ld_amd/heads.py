"""GFLHead / LDHead with mmdet's constructor arguments, state_dict keys and
method signatures (reference: mmdet/models/dense_heads/gfl_head.py:15-625,
ld_head.py:43-637, anchor_head.py:14-173, base_dense_head.py:6-59).

MI355X-native execution:
  * forward: the five FPN levels are concatenated into one (N, C, P) tensor
    and every weight-shared tower conv / GroupNorm / predictor runs as ONE
    launch over all levels (the reference runs 5 x 10 small convs);
  * loss: anchors are never materialised, targets come from two batched
    launches (targets.hip) and the whole loss_single x 5 levels, forward AND
    gradient, is the fused block of loss.hip, reading the NCHW head outputs in
    place.  No host synchronisation: the two normalisers stay on the device
    (the reference does ~90 .item()/nonzero syncs per step).
"""
import torch
import torch.nn as nn

from . import layers as Y
from . import lib as L
from . import lossblock as LB
from .cnn import ConvModule, Conv2d, Scale, bias_init_with_prob, normal_init
from .registry import (HEADS, build_anchor_generator, build_assigner,
                       build_bbox_coder, build_iou_calculator, build_loss,
                       build_sampler)

LOSS_KEYS = L.LOSS_KEYS


class Integral(nn.Module):
    """gfl_head.py:15-44: expectation of the softmax over {0..reg_max}."""

    def __init__(self, reg_max=16):
        super().__init__()
        self.reg_max = reg_max
        self.register_buffer('project',
                             torch.linspace(0, self.reg_max, self.reg_max + 1))

    def forward(self, x):
        return _IntegralFn.apply(x.reshape(-1, 4 * (self.reg_max + 1)))


class _IntegralFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        x = L.require_device(x.contiguous(), torch.float32, 'integral input')
        if x.shape[1] != 68:
            raise NotImplementedError('reg_max != 16')
        out = x.new_empty((x.shape[0], 4))
        if x.shape[0]:
            L.check(L.get_lib().ld_integral_rows(L.ptr(x), x.shape[0],
                                                 L.ptr(out),
                                                 L.stream_ptr(x.device)),
                    'ld_integral_rows')
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        gx = torch.empty_like(x)
        if x.shape[0]:
            L.check(L.get_lib().ld_integral_rows_bwd(L.ptr(x),
                                                     L.ptr(g.contiguous()),
                                                     x.shape[0], L.ptr(gx),
                                                     L.stream_ptr(x.device)),
                    'ld_integral_rows_bwd')
        return gx


class LossDict(dict):
    """dict[str, list[Tensor]] as the reference returns, plus the (8, L) device
    table it is a view of, so _parse_losses can reduce it in two launches."""
    table = None
    rows = None


def _imitation_flags(method, lw_im):
    """hp flags of the imitation region (ld_head.py:170-191,580-611).
    'finegrained': IoU > 0.5 max IoU per GT.  'fitnet': anchor centre strictly
    inside a GT.  'gibox': the GI boxes (selected after the forward; the region
    flag only matters for its weight-0 evaluation).  'decouple' adds
    ``2 * mse(x[outside], teacher_x[inside])`` -- two row sets of different
    sizes, which F.mse_loss cannot broadcast: the reference itself raises on
    that branch, so there is nothing to reproduce."""
    if method == 'decouple' and lw_im != 0.0:
        raise NotImplementedError(
            "imitation_method='decouple': ld_head.py:176-183 evaluates "
            'mse_loss(x[ng_inds], teacher_x[fg_inds]) on row sets of different '
            'sizes, which raises in the reference as well')
    return L.LD_IM_CENTER_INSIDE if method in ('fitnet', 'decouple',
                                               'gibox') else 0


class LazyScalars(dict):
    """dict of python floats backed by one device tensor; the single D2H copy
    happens on first access (the reference syncs 9 times per step).

    It stays a ``dict`` subclass because mmcv's LogBuffer.update asserts
    ``isinstance(vars, dict)``; every read path -- including the ones CPython
    serves from the raw table for plain dicts (``dict(x)``, ``{**x}``,
    ``OrderedDict(x)``, ``copy()``) -- is routed through the sync: overriding
    ``__iter__``/``keys`` takes ``dict_merge`` off its fast path."""

    def __init__(self, keys, tensor):
        super().__init__()
        self._keys, self._tensor, self._done = list(keys), tensor, False
        for k in keys:
            dict.__setitem__(self, k, None)

    def _sync(self):
        if not self._done:
            vals = self._tensor.detach().cpu().tolist()
            for k, v in zip(self._keys, vals):
                dict.__setitem__(self, k, v)
            self._done = True

    def __getitem__(self, k):
        self._sync()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._sync()
        return dict.get(self, k, default)

    def __iter__(self):
        self._sync()
        return dict.__iter__(self)

    def keys(self):
        self._sync()
        return dict.keys(self)

    def items(self):
        self._sync()
        return dict.items(self)

    def values(self):
        self._sync()
        return dict.values(self)

    def copy(self):
        self._sync()
        return dict(dict.items(self))

    def pop(self, *a):
        self._sync()
        return dict.pop(self, *a)

    def setdefault(self, k, default=None):
        self._sync()
        return dict.setdefault(self, k, default)

    def __eq__(self, other):
        self._sync()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._sync()
        return dict.__repr__(self)

    def __reduce__(self):
        self._sync()
        return (dict, (dict(dict.items(self)), ))


def _tower(convs, x3, levels):
    """A stack of ConvModules on the level-concatenated tensor.  Every layer but
    the last is told that its output feeds another conv of the stack (c8_out): a
    frozen conv + GN layer in bf16 mode then keeps only the bf16 C8 image of its
    output (the teacher's towers; cnn.ConvModule.forward3)."""
    last = len(convs) - 1
    for i, m in enumerate(convs):
        x3, _ = m.forward3(x3, levels, c8_out=i < last)
    return x3


@HEADS.register_module()
class GFLHead(nn.Module):
    """Constructor = AnchorHead.__init__ (anchor_head.py:31-96) +
    GFLHead.__init__ (gfl_head.py:76-100)."""

    def __init__(self, num_classes, in_channels, stacked_convs=4,
                 conv_cfg=None,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
                 reg_max=16, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', ratios=[1.0],
                                       octave_base_scale=8,
                                       scales_per_octave=1,
                                       strides=[8, 16, 32, 64, 128]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                                 target_means=(.0, .0, .0, .0),
                                 target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True,
                               beta=2.0, loss_weight=1.0),
                 loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
                 train_cfg=None, test_cfg=None):
        super().__init__()
        self.stacked_convs, self.conv_cfg, self.norm_cfg = (stacked_convs,
                                                            conv_cfg, norm_cfg)
        self.reg_max = reg_max
        self.in_channels, self.num_classes = in_channels, num_classes
        self.feat_channels = feat_channels
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in [
            'FocalLoss', 'GHMC', 'QualityFocalLoss'
        ]
        self.cls_out_channels = num_classes if self.use_sigmoid_cls \
            else num_classes + 1
        if self.cls_out_channels <= 0:
            raise ValueError(f'num_classes={num_classes} is too small')
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            self.sampler = build_sampler(dict(type='PseudoSampler'),
                                         context=self)
        self.sampling = False
        self.fp16_enabled = False
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self._init_layers()
        self.integral = Integral(self.reg_max)
        self.loss_dfl = build_loss(loss_dfl)
        # see SGDTrainer.step: only set for the duration of a train step
        self.unit_upstream = False

    def _init_layers(self):
        """gfl_head.py:102-133."""
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            self.reg_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        assert self.num_anchors == 1, 'anchor free version'
        self.gfl_cls = Conv2d(self.feat_channels, self.cls_out_channels, 3,
                              padding=1)
        self.gfl_reg = Conv2d(self.feat_channels, 4 * (self.reg_max + 1), 3,
                              padding=1)
        self.scales = nn.ModuleList(
            [Scale(1.0) for _ in self.anchor_generator.strides])

    def init_weights(self):
        """gfl_head.py:135-143."""
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.gfl_cls, std=0.01, bias=bias_cls)
        normal_init(self.gfl_reg, std=0.01)

    # ------------------------------------------------------------ forward --
    def forward(self, feats):
        """feats: tuple of per-level (N, C, H, W) -> (cls_scores, bbox_preds)
        lists (gfl_head.py:145-183), all levels in one launch per layer."""
        assert len(feats) == len(self.scales)
        x3, levels = self._pack(feats)
        cls_feat = reg_feat = x3
        cls_feat = _tower(self.cls_convs, cls_feat, levels)
        reg_feat = _tower(self.reg_convs, reg_feat, levels)
        cls3, _ = self.gfl_cls.forward3(cls_feat, levels)
        reg3, _ = self.gfl_reg.forward3(reg_feat, levels)
        scales = torch.stack([s.scale for s in self.scales])
        reg3 = Y.scale_levels(reg3, scales, levels)
        return Y.split_levels(cls3, levels), Y.split_levels(reg3, levels)

    _wants_packed_feats = False  # LDHead / LDv2Head: their loss reads the features

    def _pack(self, feats):
        """The level-concatenated head input; remembered for ``_loss_feats``
        while a gradient is wanted."""
        x3, levels = Y.pack_levels(feats)
        self._packed = (x3, levels) if self._wants_packed_feats and \
            torch.is_grad_enabled() and x3.requires_grad and \
            len(feats) > 1 else None
        return x3, levels

    def _loss_feats(self, x):
        """The neck features the LD loss block reads (ld_head.py:284-375: ``x``
        for the imitation term): the level VIEWS of the packed head input --
        the same values as ``x``, but the imitation gradient then re-enters the
        graph at the packed tensor, where the first tower conv's data gradient
        sums it in its epilogue (layers.fan_*), instead of five per-level
        elementwise adds at the neck outputs."""
        pk, self._packed = getattr(self, '_packed', None), None
        return x if pk is None else Y.split_levels(*pk)

    def forward_single(self, x, scale):
        raise NotImplementedError(
            'use forward(feats): all levels run in one launch per layer')

    def anchor_center(self, anchors):
        """gfl_head.py:185-194."""
        cx = (anchors[..., 2] + anchors[..., 0]) / 2
        cy = (anchors[..., 3] + anchors[..., 1]) / 2
        return torch.stack([cx, cy], dim=-1)

    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        """anchor_head.py:145-173 (API compatibility; the loss never calls
        it)."""
        num_imgs = len(img_metas)
        multi_level_anchors = self.anchor_generator.grid_anchors(
            featmap_sizes, device)
        anchor_list = [multi_level_anchors for _ in range(num_imgs)]
        valid_flag_list = []
        for meta in img_metas:
            valid_flag_list.append(
                self.anchor_generator.valid_flags(featmap_sizes,
                                                  meta['pad_shape'], device))
        return anchor_list, valid_flag_list

    # --------------------------------------------------------------- loss --
    def _hp(self, **over):
        kw = dict(
            num_classes=self.num_classes, reg_max=self.reg_max,
            topk=self.assigner.topk, feat_channels=self.feat_channels,
            lw_cls=self.loss_cls.loss_weight,
            qfl_beta=getattr(self.loss_cls, 'beta', 2.0),
            lw_bbox=self.loss_bbox.loss_weight,
            giou_eps=getattr(self.loss_bbox, 'eps', 1e-6),
            lw_dfl=self.loss_dfl.loss_weight, lw_ld=0.0, T_ld=1.0,
            lw_ld_vlr=0.0, T_ld_vlr=1.0, lw_kd=0.0, T_kd=1.0, lw_im=0.0)
        kw.update(over)
        return LB.make_hp(**kw)

    def _check_loss_cfg(self):
        from .losses import GIoULoss, QualityFocalLoss
        if not isinstance(self.loss_cls, QualityFocalLoss) or \
                not isinstance(self.loss_bbox, GIoULoss):
            raise NotImplementedError(
                'the fused loss block implements QualityFocalLoss + GIoULoss '
                f'(got {type(self.loss_cls).__name__}, '
                f'{type(self.loss_bbox).__name__})')
        if self.train_cfg.get('allowed_border', -1) >= 0:
            raise NotImplementedError('allowed_border >= 0')
        if self.train_cfg.get('pos_weight', -1) > 0:
            raise NotImplementedError('pos_weight > 0')

    def get_targets_batched(self, featmap_sizes, img_metas, gt_bboxes,
                            gt_labels, hp, device):
        """AnchorHead.get_anchors + LDHead.get_targets for the whole batch in
        two launches (ld_head.py:377-577)."""
        strides = [s[0] for s in self.anchor_generator.strides]
        if not self.anchor_generator.single_square:
            # the implicit-anchor kernels build the one square anchor of a cell
            # from an INTEGER scale (octave_base_scale * stride)
            raise NotImplementedError(
                f'{type(self).__name__}: one anchor per cell with a non-integer '
                'scale or ratio != 1 (anchor_generator.py:78-98); the '
                'single-anchor target kernels take ratios=[1.0] with an integer '
                'octave_base_scale')
        if gt_labels is None:
            gt_labels = [b.new_zeros(b.shape[0], dtype=torch.long)
                         for b in gt_bboxes]
        return LB.atss_targets(featmap_sizes, strides, img_metas, gt_bboxes,
                               gt_labels, hp, device,
                               self.anchor_generator.anchor_scale)

    @staticmethod
    def _norm_reducer():
        """Cross-rank mean of (num_total_pos, sum weight_targets): ONE device
        side all-reduce instead of the reference's two reduce_mean(...).item()
        host syncs (ld_head.py:338-341,362-363)."""
        import torch.distributed as dist
        from .train import _diag_skip, collectives_on
        if not collectives_on() or _diag_skip('norm'):
            return None
        ws = float(dist.get_world_size())

        def _r(norm):
            dist.all_reduce(norm)
            norm.div_(ws)

        return _r

    def _loss_dict(self, table, keys=LOSS_KEYS):
        d = LossDict((k, [table[i, l] for l in range(table.shape[1])])
                     for i, k in enumerate(LOSS_KEYS) if k in keys)
        d.table, d.rows = table, [i for i, k in enumerate(LOSS_KEYS)
                                  if k in keys]
        return d

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas,
             gt_bboxes_ignore=None):
        """gfl_head.py:269-352 (plain GFL: QFL + GIoU + DFL)."""
        self._check_loss_cfg()
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        device = cls_scores[0].device
        hp = self._hp()
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, hp, device)
        # no teacher, no features: feed the student's own (detached) outputs;
        # every distillation weight is zero
        dummy_x = [c.detach() for c in cls_scores]
        hp.feat_channels = cls_scores[0].shape[1]
        teacher = ([c.detach() for c in cls_scores],
                   [b.detach() for b in bbox_preds], dummy_x)
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream,
                                        *cls_scores, *bbox_preds, *dummy_x)
        return self._loss_dict(table, ('loss_cls', 'loss_bbox', 'loss_dfl'))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None,
                      gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        """base_dense_head.py:20-59."""
        outs = self(x)
        losses = self.loss(*outs, gt_bboxes, gt_labels, img_metas,
                           gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is not None:
            raise NotImplementedError('get_bboxes (inference) is a "next" row '
                                      'of SURVEY.md section 8f')
        return losses

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None,
                   rescale=False, with_nms=True):
        """anchor_head.py:497-589 + gfl_head.py:354-451 + multiclass_nms, one
        C-ABI call for the whole batch (ld_get_bboxes).  Returns, per image,
        ``(det_bboxes (k, 5), det_labels (k,))`` like the reference."""
        return self._get_bboxes(cls_scores, bbox_preds, img_metas, cfg,
                                rescale, with_nms, prob=False)

    def _get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg, rescale,
                    with_nms, prob, centernesses=None, points=False,
                    strides=None, num_base=1):
        cfg = self.test_cfg if cfg is None else cfg
        if cfg is None:
            raise ValueError('get_bboxes needs a test_cfg')
        nms = cfg['nms'] if isinstance(cfg, dict) else cfg.nms
        nms_type = nms.get('type', 'nms')
        if nms_type not in ('nms', 'voting_cluster_diounms'):
            raise NotImplementedError(
                f"nms type {nms_type!r}: the reference's multiclass_nms knows "
                "'nms' and 'voting_cluster_diounms' (bbox_nms.py:141-188)")
        get = cfg.get if hasattr(cfg, 'get') else lambda k, d=None: cfg[k]
        if get('min_bbox_size', 0) not in (0, -1):
            raise NotImplementedError('min_bbox_size > 0')
        if centernesses is not None and nms_type != 'nms' and with_nms:
            raise NotImplementedError('score voting with centerness factors')
        if not with_nms:
            nms_type = 'nms'  # the nms config is not consulted
        strides = [s[0] if isinstance(s, (tuple, list)) else s
                   for s in (strides or self.anchor_generator.strides)]
        N = cls_scores[0].shape[0]
        shapes = [img_metas[i]['img_shape'] for i in range(N)]
        sfs = [img_metas[i]['scale_factor'] for i in range(N)] if rescale \
            else None
        return LB.get_bboxes(
            [c.detach() for c in cls_scores], [b.detach() for b in bbox_preds],
            strides, shapes, sfs, nms_pre=get('nms_pre', -1),
            score_thr=get('score_thr'), iou_thr=nms['iou_threshold'],
            max_per_img=get('max_per_img'), num_classes=self.cls_out_channels,
            reg_max=self.reg_max, voting=nms_type == 'voting_cluster_diounms',
            prob=prob, centernesses=None if centernesses is None else
            [c.detach() for c in centernesses], points=points,
            num_base=num_base, with_nms=with_nms)


@HEADS.register_module()
class LDHead(GFLHead):
    """ld_head.py:43-637."""
    _wants_packed_feats = True

    def __init__(self, num_classes, in_channels,
                 loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                              loss_weight=0.25, T=10),
                 loss_ld_vlr=dict(type='KnowledgeDistillationKLDivLoss',
                                  loss_weight=0.25, T=10),
                 loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                              loss_weight=10, T=2),
                 loss_im=dict(type='IMLoss', loss_weight=0),
                 imitation_method='gibox', **kwargs):
        super().__init__(num_classes, in_channels, **kwargs)
        assert imitation_method in ['gibox', 'finegrained', 'fitnet',
                                    'decouple']
        self.imitation_method = imitation_method
        self.loss_im = build_loss(loss_im)
        self.loss_ld = build_loss(loss_ld)
        self.loss_ld_vlr = build_loss(loss_ld_vlr)
        self.loss_kd = build_loss(loss_kd)
        self.iou_calculator = build_iou_calculator(dict(type='BboxOverlaps2D'))
        # d(total)/d(loss_k) == 1 is promised by BaseDetector._parse_losses;
        # the train engine sets this so the backward reuses the gradient the
        # fused forward launch already produced
        self.unit_upstream = False

    def _imitation_flags(self, lw_im):
        return _imitation_flags(self.imitation_method, lw_im)

    def forward_train(self, x, out_teacher, teacher_x, img_metas, gt_bboxes,
                      gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """ld_head.py:73-114."""
        outs = self(x)
        if gt_labels is None:
            raise NotImplementedError('LDHead needs gt_labels')
        losses = self.loss(*outs, gt_bboxes, gt_labels, out_teacher,
                           self._loss_feats(x), teacher_x, img_metas,
                           gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is not None:
            raise NotImplementedError('get_bboxes (inference) is a "next" row '
                                      'of SURVEY.md section 8f')
        return losses

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, soft_teacher,
             x, teacher_x, img_metas, gt_bboxes_ignore=None):
        """ld_head.py:284-375 -> dict of 8 lists of per-level scalars."""
        self._check_loss_cfg()
        lw_im = float(self.loss_im.loss_weight)
        im_flags = self._imitation_flags(lw_im)
        if x[0].shape[1] != 256:
            raise ValueError('LDHead hard-codes 256 feature channels '
                             '(ld_head.py:153-154)')
        soft_label, soft_target = soft_teacher
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        assert len(sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        hp = self._hp(lw_ld=self.loss_ld.loss_weight, T_ld=self.loss_ld.T,
                      lw_ld_vlr=self.loss_ld_vlr.loss_weight,
                      T_ld_vlr=self.loss_ld_vlr.T,
                      lw_kd=self.loss_kd.loss_weight, T_kd=self.loss_kd.T,
                      lw_im=lw_im, flags=im_flags)
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, hp, device)
        teacher = ([t.detach() for t in soft_label],
                   [t.detach() for t in soft_target],
                   [t.detach() for t in teacher_x])
        if self.imitation_method == 'gibox' and lw_im != 0.0:
            targets = LB.gi_region(hp, targets,
                                   [c.detach() for c in cls_scores],
                                   [b.detach() for b in bbox_preds],
                                   teacher[0], teacher[1])
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream, *cls_scores,
                                        *bbox_preds, *x)
        self.last_targets = targets
        return self._loss_dict(table)


ATSS_LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_ld', 'loss_ld_neg',
                  'loss_cls_kd', 'loss_centerness']
# rows of the fused block's (8, L) table that carry them (LD_LOSS_ATSS)
_ATSS_ROWS = [0, 1, 3, 4, 5, 6]


@HEADS.register_module()
class ATSSGFLHead(GFLHead):
    """atss_gfl_head.py:52-185: the ATSS head with a general-distribution box
    branch -- GFLHead's two towers, ``atss_cls`` / ``atss_reg`` /
    ``atss_centerness`` output convs (state_dict names of the reference),
    FocalLoss + centerness-weighted GIoU + centerness BCE.  forward returns
    (cls_scores, bbox_preds, centernesses)."""

    def __init__(self, num_classes, in_channels, stacked_convs=4,
                 conv_cfg=None,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 loss_centerness=dict(type='CrossEntropyLoss',
                                      use_sigmoid=True, loss_weight=1.0),
                 reg_max=16, **kwargs):
        kwargs.setdefault('loss_cls', dict(type='FocalLoss', use_sigmoid=True,
                                           gamma=2.0, alpha=0.25,
                                           loss_weight=1.0))
        kwargs.setdefault('bbox_coder', dict(
            type='DeltaXYWHBBoxCoder', target_means=(.0, .0, .0, .0),
            target_stds=(1.0, 1.0, 1.0, 1.0)))
        super().__init__(num_classes, in_channels, stacked_convs=stacked_convs,
                         conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                         loss_dfl=dict(type='DistributionFocalLoss',
                                       loss_weight=0.0),
                         reg_max=reg_max, **kwargs)
        self.loss_centerness = build_loss(loss_centerness)

    def _init_layers(self):
        """atss_gfl_head.py:90-125."""
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            self.reg_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        assert self.num_anchors == 1, 'one square anchor per position'
        self.atss_cls = Conv2d(self.feat_channels,
                               self.num_anchors * self.cls_out_channels, 3,
                               padding=1)
        self.atss_reg = Conv2d(self.feat_channels, 4 * (self.reg_max + 1), 3,
                               padding=1)
        self.atss_centerness = Conv2d(self.feat_channels, self.num_anchors, 3,
                                      padding=1)
        self.scales = nn.ModuleList(
            [Scale(1.0) for _ in self.anchor_generator.strides])

    def init_weights(self):
        """atss_gfl_head.py:127-137."""
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        normal_init(self.atss_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.atss_reg, std=0.01)
        normal_init(self.atss_centerness, std=0.01)

    def forward(self, feats):
        """atss_gfl_head.py:139-183, all levels in one launch per layer."""
        assert len(feats) == len(self.scales)
        x3, levels = self._pack(feats)
        cls_feat = reg_feat = x3
        cls_feat = _tower(self.cls_convs, cls_feat, levels)
        reg_feat = _tower(self.reg_convs, reg_feat, levels)
        cls3, _ = self.atss_cls.forward3(cls_feat, levels)
        reg3, _ = self.atss_reg.forward3(reg_feat, levels)
        ctr3, _ = self.atss_centerness.forward3(reg_feat, levels)
        scales = torch.stack([s.scale for s in self.scales])
        reg3 = Y.scale_levels(reg3, scales, levels)
        return (Y.split_levels(cls3, levels), Y.split_levels(reg3, levels),
                Y.split_levels(ctr3, levels))

    def _check_loss_cfg(self):
        from .losses import CrossEntropyLoss, FocalLoss, GIoULoss
        if not isinstance(self.loss_cls, FocalLoss) or \
                not isinstance(self.loss_bbox, GIoULoss) or \
                not isinstance(self.loss_centerness, CrossEntropyLoss) or \
                not self.loss_centerness.use_sigmoid:
            raise NotImplementedError(
                'the fused ATSS loss block implements FocalLoss + GIoULoss + '
                'sigmoid CrossEntropyLoss centerness')
        if self.loss_cls.gamma != 2.0:
            raise NotImplementedError('FocalLoss gamma != 2')
        if self.train_cfg.get('allowed_border', -1) >= 0:
            raise NotImplementedError('allowed_border >= 0')
        if self.train_cfg.get('pos_weight', -1) > 0:
            raise NotImplementedError('pos_weight > 0')

    def _hp(self, **over):
        kw = dict(lw_dfl=0.0, lw_ctr=self.loss_centerness.loss_weight,
                  focal_alpha=self.loss_cls.alpha, qfl_beta=2.0,
                  flags=L.LD_LOSS_ATSS)
        kw.update(over)
        return super()._hp(**kw)

    def _atss_loss_dict(self, table, keys):
        d = LossDict((k, [table[r, l] for l in range(table.shape[1])])
                     for k, r in zip(ATSS_LOSS_KEYS, _ATSS_ROWS) if k in keys)
        d.table = table
        d.rows = [r for k, r in zip(ATSS_LOSS_KEYS, _ATSS_ROWS) if k in keys]
        return d

    def loss(self, cls_scores, bbox_preds, centernesses, gt_bboxes, gt_labels,
             img_metas, gt_bboxes_ignore=None):
        """atss_gfl_head.py:187-310: loss_cls, loss_bbox, loss_centerness."""
        self._check_loss_cfg()
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        device = cls_scores[0].device
        hp = self._hp()
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, hp, device)
        dummy_x = [c.detach() for c in cls_scores]
        hp.feat_channels = cls_scores[0].shape[1]
        teacher = ([c.detach() for c in cls_scores],
                   [b.detach() for b in bbox_preds], dummy_x)
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream, *cls_scores,
                                        *bbox_preds, *dummy_x, *centernesses)
        return self._atss_loss_dict(table, ('loss_cls', 'loss_bbox',
                                            'loss_centerness'))

    def get_bboxes(self, cls_scores, bbox_preds, centernesses, img_metas,
                   cfg=None, rescale=False, with_nms=True):
        """atss_gfl_head.py:420-575: GFLHead's pipeline with the top-k key
        max_c score_c * sigmoid(centerness) and the centerness as
        multiclass_nms' score factor (applied after the threshold test)."""
        return self._get_bboxes(cls_scores, bbox_preds, img_metas, cfg,
                                rescale, with_nms, prob=False,
                                centernesses=centernesses)


@HEADS.register_module()
class LDATSSHead(ATSSGFLHead):
    """ld_atss.py:13-250: localization distillation on the ATSS-GFL head --
    LD on the positives weighted by the max class score, 0.15 x LD on the
    valuable localisation region, KD on the positives' class logits.  The
    detector calls it with output_feature=False:
    forward_train(x, out_teacher, img_metas, ...)."""

    def __init__(self, num_classes, in_channels,
                 loss_ld=dict(type='LocalizationDistillationLoss',
                              loss_weight=0.25, T=10),
                 loss_kd=None, **kwargs):
        super().__init__(num_classes, in_channels, **kwargs)
        self.loss_ld = build_loss(loss_ld)
        self.loss_kd = build_loss(loss_kd)

    def forward_train(self, x, out_teacher, img_metas, gt_bboxes,
                      gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """ld_atss.py:252-290."""
        outs = self(x)
        if gt_labels is None:
            raise NotImplementedError('LDATSSHead needs gt_labels')
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*outs, gt_bboxes, gt_labels, out_teacher, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    def loss(self, cls_scores, bbox_preds, centernesses, gt_bboxes, gt_labels,
             soft_target, img_metas, gt_bboxes_ignore=None):
        """ld_atss.py:168-250 -> the six keys of ATSS_LOSS_KEYS."""
        self._check_loss_cfg()
        soft_labels, soft_corners = soft_target[0], soft_target[1]
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        assert len(sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        # loss_ld_neg = 0.15 * loss_ld(..., avg_factor=4) on the VLR region
        # (ld_atss.py:148-159); the block's VLR term is lw_ld_vlr * sum / 16
        hp = self._hp(lw_ld=self.loss_ld.loss_weight, T_ld=self.loss_ld.T,
                      lw_ld_vlr=0.15 * 4.0 * self.loss_ld.loss_weight,
                      T_ld_vlr=self.loss_ld.T,
                      lw_kd=self.loss_kd.loss_weight, T_kd=self.loss_kd.T)
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, hp, device)
        dummy_x = [c.detach() for c in cls_scores]
        hp.feat_channels = cls_scores[0].shape[1]
        teacher = ([t.detach() for t in soft_labels],
                   [t.detach() for t in soft_corners], dummy_x)
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream, *cls_scores,
                                        *bbox_preds, *dummy_x, *centernesses)
        self.last_targets = targets
        return self._atss_loss_dict(table, ATSS_LOSS_KEYS)


INF = 1e8


@HEADS.register_module()
class FCOSGFLHead(nn.Module):
    """fcos_gfl_head.py:52-346 over anchor_free_head.py:15-130: the anchor-free
    FCOS head with a general-distribution box branch.  Parameters
    ``cls_convs / reg_convs / conv_cls / conv_reg / conv_centerness / scales``
    as in the reference; forward returns (cls_scores, bbox_preds,
    centernesses); points are (x, y) * stride + stride // 2."""

    def __init__(self, num_classes, in_channels, feat_channels=256,
                 stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 dcn_on_last_conv=False, conv_bias='auto',
                 regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512),
                                 (512, INF)),
                 center_sampling=False, center_sample_radius=1.5,
                 norm_on_bbox=False, centerness_on_reg=False,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                               alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='IoULoss', loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss',
                                      use_sigmoid=True, loss_weight=1.0),
                 reg_max=16, conv_cfg=None,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 train_cfg=None, test_cfg=None):
        super().__init__()
        if dcn_on_last_conv:
            raise NotImplementedError('dcn_on_last_conv')
        if norm_on_bbox:
            raise NotImplementedError('norm_on_bbox=True')
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.stacked_convs, self.strides = stacked_convs, list(strides)
        self.dcn_on_last_conv, self.conv_bias = dcn_on_last_conv, conv_bias
        self.regress_ranges = regress_ranges
        self.center_sampling = center_sampling
        self.center_sample_radius = center_sample_radius
        self.norm_on_bbox, self.centerness_on_reg = (norm_on_bbox,
                                                     centerness_on_reg)
        self.reg_max = reg_max
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_centerness = build_loss(loss_centerness)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.fp16_enabled = False
        self.unit_upstream = False
        self._init_layers()
        # after the layers, like the reference (state_dict key order)
        self.integral = Integral(reg_max)

    def _init_layers(self):
        """fcos_gfl_head.py:134-165."""
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            self.reg_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        self.conv_cls = Conv2d(self.feat_channels, self.cls_out_channels, 3,
                               padding=1)
        self.conv_reg = Conv2d(self.feat_channels, 4 * (self.reg_max + 1), 3,
                               padding=1)
        self.conv_centerness = Conv2d(self.feat_channels, 1, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.strides])

    def init_weights(self):
        """fcos_gfl_head.py:167-176."""
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        normal_init(self.conv_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.conv_reg, std=0.01)
        normal_init(self.conv_centerness, std=0.01)

    def forward(self, feats):
        """fcos_gfl_head.py:178-224, all levels in one launch per layer."""
        assert len(feats) == len(self.scales)
        x3, levels = Y.pack_levels(feats)
        cls_feat = reg_feat = x3
        cls_feat = _tower(self.cls_convs, cls_feat, levels)
        reg_feat = _tower(self.reg_convs, reg_feat, levels)
        cls3, _ = self.conv_cls.forward3(cls_feat, levels)
        reg3, _ = self.conv_reg.forward3(reg_feat, levels)
        ctr3, _ = self.conv_centerness.forward3(reg_feat, levels)
        scales = torch.stack([s.scale for s in self.scales])
        reg3 = Y.scale_levels(reg3, scales, levels)
        return (Y.split_levels(cls3, levels), Y.split_levels(reg3, levels),
                Y.split_levels(ctr3, levels))

    # ---------------------------------------------------------------- loss --
    _norm_reducer = staticmethod(GFLHead._norm_reducer)

    def _check_loss_cfg(self):
        from .losses import CrossEntropyLoss, FocalLoss, GIoULoss
        if not isinstance(self.loss_cls, FocalLoss) or \
                not isinstance(self.loss_bbox, GIoULoss) or \
                not isinstance(self.loss_centerness, CrossEntropyLoss) or \
                not self.loss_centerness.use_sigmoid:
            raise NotImplementedError(
                'the fused FCOS loss block implements FocalLoss + GIoULoss + '
                'sigmoid CrossEntropyLoss centerness')
        if self.loss_cls.gamma != 2.0:
            raise NotImplementedError('FocalLoss gamma != 2')

    def _hp(self, **over):
        kw = dict(num_classes=self.num_classes, reg_max=self.reg_max, topk=9,
                  feat_channels=self.feat_channels,
                  lw_cls=self.loss_cls.loss_weight, qfl_beta=2.0,
                  lw_bbox=self.loss_bbox.loss_weight,
                  giou_eps=getattr(self.loss_bbox, 'eps', 1e-6), lw_dfl=0.0,
                  lw_ld=0.0, T_ld=1.0, lw_ld_vlr=0.0, T_ld_vlr=1.0, lw_kd=0.0,
                  T_kd=1.0, lw_im=0.0,
                  lw_ctr=self.loss_centerness.loss_weight,
                  focal_alpha=self.loss_cls.alpha,
                  flags=L.LD_LOSS_ATSS | L.LD_LOSS_FCOS)
        kw.update(over)
        return LB.make_hp(**kw)

    def get_targets_batched(self, featmap_sizes, gt_bboxes, gt_labels, device):
        """get_points + get_targets (ld_fcos_head.py:261-414) for the whole
        batch in one launch."""
        return LB.fcos_targets(featmap_sizes, self.strides, gt_bboxes,
                               gt_labels, self.num_classes,
                               self.regress_ranges, self.center_sampling,
                               self.center_sample_radius, device)

    def _run_block(self, hp, cls_scores, bbox_preds, centernesses, gt_bboxes,
                   gt_labels, t_cls, t_reg, keys):
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        assert len(sizes) == len(self.strides)
        device = cls_scores[0].device
        targets = self.get_targets_batched(sizes, gt_bboxes, gt_labels, device)
        dummy_x = [c.detach() for c in cls_scores]
        hp.feat_channels = cls_scores[0].shape[1]
        teacher = ([t.detach() for t in t_cls], [t.detach() for t in t_reg],
                   dummy_x)
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream, *cls_scores,
                                        *bbox_preds, *dummy_x, *centernesses)
        self.last_targets = targets
        d = LossDict((k, [table[r, l] for l in range(table.shape[1])])
                     for k, r in zip(ATSS_LOSS_KEYS, _ATSS_ROWS) if k in keys)
        d.table = table
        d.rows = [r for k, r in zip(ATSS_LOSS_KEYS, _ATSS_ROWS) if k in keys]
        return d

    def loss(self, cls_scores, bbox_preds, centernesses, gt_bboxes, gt_labels,
             img_metas, gt_bboxes_ignore=None):
        """fcos_gfl_head.py:276-345: loss_cls, loss_bbox, loss_centerness."""
        self._check_loss_cfg()
        return self._run_block(self._hp(), cls_scores, bbox_preds,
                               centernesses, gt_bboxes, gt_labels, cls_scores,
                               bbox_preds, ('loss_cls', 'loss_bbox',
                                            'loss_centerness'))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None,
                      gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*self(x), gt_bboxes, gt_labels, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    def get_bboxes(self, cls_scores, bbox_preds, centernesses, img_metas,
                   cfg=None, rescale=False, with_nms=True):
        """fcos_gfl_head.py:347-546: as ATSSGFLHead.get_bboxes, decoded about
        the FCOS points (x, y) * stride + stride // 2."""
        return GFLHead._get_bboxes(self, cls_scores, bbox_preds, img_metas,
                                   cfg, rescale, with_nms, prob=False,
                                   centernesses=centernesses, points=True,
                                   strides=self.strides)


@HEADS.register_module()
class LDFCOSHead(FCOSGFLHead):
    """ld_fcos_head.py:13-445: localization distillation on the FCOS-GFL head:
    LD on the positives weighted by the max class score, 0.25 x LD on the
    "remain" points (inside a gt box, assigned to none) weighted by the
    student's max class score, KD on the positives' class logits."""

    def __init__(self, num_classes, in_channels,
                 loss_ld=dict(type='LocalizationDistillationLoss',
                              loss_weight=0.25, T=10),
                 loss_kd=None, **kwargs):
        super().__init__(num_classes, in_channels, **kwargs)
        self.loss_ld = build_loss(loss_ld)
        self.loss_kd = build_loss(loss_kd)

    def forward_train(self, x, out_teacher, img_metas, gt_bboxes,
                      gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """ld_fcos_head.py:219-259."""
        if gt_labels is None:
            raise NotImplementedError('LDFCOSHead needs gt_labels')
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*self(x), gt_bboxes, gt_labels, out_teacher,
                         img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    def loss(self, cls_scores, bbox_preds, centernesses, gt_bboxes, gt_labels,
             out_teacher, img_metas, gt_bboxes_ignore=None):
        """ld_fcos_head.py:138-217 -> the six keys of ATSS_LOSS_KEYS."""
        self._check_loss_cfg()
        soft_labels, soft_targets = out_teacher[0], out_teacher[1]
        # loss_ld_neg = 0.25 * loss_ld(..., avg_factor=4) (ld_fcos_head.py:
        # 125-129); the block's VLR term is lw_ld_vlr * sum / 16
        hp = self._hp(lw_ld=self.loss_ld.loss_weight, T_ld=self.loss_ld.T,
                      lw_ld_vlr=0.25 * 4.0 * self.loss_ld.loss_weight,
                      T_ld_vlr=self.loss_ld.T,
                      lw_kd=self.loss_kd.loss_weight, T_kd=self.loss_kd.T)
        return self._run_block(hp, cls_scores, bbox_preds, centernesses,
                               gt_bboxes, gt_labels, soft_labels, soft_targets,
                               ATSS_LOSS_KEYS)


RETINA_LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_ld', 'loss_ld_vlr',
                    'loss_cls_kd']
# rows of the fused block's (8, L) table that carry them (LD_LOSS_RETINA)
_RETINA_ROWS = [0, 1, 3, 4, 5]


@HEADS.register_module()
class RetinaGFLHead(nn.Module):
    """retina_gfl_head.py:50-330 over anchor_head.py:14-173: the RetinaNet head
    (ratios x scales anchors per cell, conv + ReLU towers without a norm
    layer) with a general-distribution box branch.  ``atss_cls`` /
    ``atss_reg`` predictor names as in the reference; forward returns
    (cls_scores (N, B * C, H, W), bbox_preds (N, B * 68, H, W)) lists.

    Loss execution: the B anchors of a cell are B pseudo-images of the fused
    one-anchor-per-cell loss block (an (N, B * C, H, W) map IS the (N * B, C,
    H, W) map of them), targets come from ld_retina_targets."""

    def __init__(self, num_classes, in_channels, stacked_convs=4,
                 conv_cfg=None, norm_cfg=None, reg_max=16, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator',
                                       octave_base_scale=4,
                                       scales_per_octave=3,
                                       ratios=[0.5, 1.0, 2.0],
                                       strides=[8, 16, 32, 64, 128]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                                 target_means=(.0, .0, .0, .0),
                                 target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True,
                               loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0,
                                loss_weight=1.0),
                 train_cfg=None, test_cfg=None):
        super().__init__()
        self.stacked_convs, self.conv_cfg, self.norm_cfg = (stacked_convs,
                                                            conv_cfg, norm_cfg)
        self.reg_max = reg_max
        self.in_channels, self.num_classes = in_channels, num_classes
        self.feat_channels = feat_channels
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in [
            'FocalLoss', 'GHMC', 'QualityFocalLoss']
        self.cls_out_channels = num_classes if self.use_sigmoid_cls \
            else num_classes + 1
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            self.sampler = build_sampler(dict(type='PseudoSampler'),
                                         context=self)
        self.fp16_enabled = False
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self._init_layers()
        self.integral = Integral(self.reg_max)
        self.unit_upstream = False

    def _init_layers(self):
        """retina_gfl_head.py:231-264."""
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            self.reg_convs.append(
                ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                           conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        self.atss_cls = Conv2d(self.feat_channels,
                               self.num_anchors * self.cls_out_channels, 3,
                               padding=1)
        self.atss_reg = Conv2d(self.feat_channels,
                               self.num_anchors * (self.reg_max + 1) * 4, 3,
                               padding=1)

    def init_weights(self):
        """retina_gfl_head.py:266-274."""
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        normal_init(self.atss_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.atss_reg, std=0.01)

    def forward(self, feats):
        """retina_gfl_head.py:276-299, all levels in one launch per layer."""
        x3, levels = Y.pack_levels(feats)
        cls_feat = reg_feat = x3
        cls_feat = _tower(self.cls_convs, cls_feat, levels)
        reg_feat = _tower(self.reg_convs, reg_feat, levels)
        cls3, _ = self.atss_cls.forward3(cls_feat, levels)
        reg3, _ = self.atss_reg.forward3(reg_feat, levels)
        return Y.split_levels(cls3, levels), Y.split_levels(reg3, levels)

    anchor_center = GFLHead.anchor_center
    get_anchors = GFLHead.get_anchors

    # ---------------------------------------------------------------- loss --
    def _check_loss_cfg(self):
        from .losses import FocalLoss, GIoULoss
        if not isinstance(self.loss_cls, FocalLoss) or \
                not isinstance(self.loss_bbox, GIoULoss) or \
                not self.reg_decoded_bbox:
            raise NotImplementedError(
                'the fused RetinaGFL loss block implements FocalLoss + '
                'GIoULoss on decoded boxes (reg_decoded_bbox=True)')
        if self.loss_cls.gamma != 2.0:
            raise NotImplementedError('FocalLoss gamma != 2')
        if self.train_cfg.get('allowed_border', -1) >= 0:
            raise NotImplementedError('allowed_border >= 0')
        if self.train_cfg.get('pos_weight', -1) > 0:
            raise NotImplementedError('pos_weight > 0')
        if type(self.assigner).__name__ != 'MaxIoUAssigner':
            raise NotImplementedError(
                f'{type(self.assigner).__name__}: the RetinaGFL targets kernel '
                'implements MaxIoUAssigner')

    def _hp(self, **over):
        kw = dict(num_classes=self.num_classes, reg_max=self.reg_max, topk=9,
                  feat_channels=self.feat_channels,
                  lw_cls=self.loss_cls.loss_weight, qfl_beta=2.0,
                  lw_bbox=self.loss_bbox.loss_weight,
                  giou_eps=getattr(self.loss_bbox, 'eps', 1e-6), lw_dfl=0.0,
                  lw_ld=0.0, T_ld=1.0, lw_ld_vlr=0.0, T_ld_vlr=1.0, lw_kd=0.0,
                  T_kd=1.0, lw_im=0.0, focal_alpha=self.loss_cls.alpha,
                  flags=L.LD_LOSS_RETINA)
        kw.update(over)
        return LB.make_hp(**kw)

    def get_targets_batched(self, featmap_sizes, img_metas, gt_bboxes,
                            gt_labels, device, want_gt_inds=False):
        """get_anchors + get_targets (ld_retina.py:364-470) for the whole
        batch: three launches."""
        strides = [s[0] for s in self.anchor_generator.strides]
        if gt_labels is None:
            gt_labels = [b.new_zeros(b.shape[0], dtype=torch.long)
                         for b in gt_bboxes]
        anchors = [self.anchor_generator.grid_anchors_flat(featmap_sizes,
                                                           device)]
        return LB.retina_targets(featmap_sizes, strides, img_metas, gt_bboxes,
                                 gt_labels, anchors, self.num_anchors,
                                 self.assigner, self.num_classes, device,
                                 want_gt_inds=want_gt_inds)

    def _pseudo(self, maps, channels):
        """(N, B * C, H, W) -> the (N * B, C, H, W) view of the same memory."""
        B = self.num_anchors
        return [t.view(t.shape[0] * B, channels, t.shape[2], t.shape[3])
                for t in maps]

    def _run_block(self, hp, cls_scores, bbox_preds, gt_bboxes, gt_labels,
                   img_metas, t_cls, t_reg, keys):
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        assert len(sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, device)
        C_, R4 = self.cls_out_channels, 4 * (self.reg_max + 1)
        cls_p, reg_p = self._pseudo(cls_scores, C_), self._pseudo(bbox_preds,
                                                                  R4)
        t_cls_p = self._pseudo([t.detach() for t in t_cls], C_)
        t_reg_p = self._pseudo([t.detach() for t in t_reg], R4)
        dummy_x = [c.detach() for c in cls_p]
        hp.feat_channels = C_
        # num_total_samples is the LOCAL count (ld_retina.py:228-229): no
        # cross-rank reduction of the normaliser
        table, _ = LB.LDLossBlock.apply(hp, targets, (t_cls_p, t_reg_p,
                                                      dummy_x), None,
                                        self.unit_upstream, *cls_p, *reg_p,
                                        *dummy_x)
        self.last_targets = targets
        d = LossDict((k, [table[r, l] for l in range(table.shape[1])])
                     for k, r in zip(RETINA_LOSS_KEYS, _RETINA_ROWS)
                     if k in keys)
        d.table = table
        d.rows = [r for k, r in zip(RETINA_LOSS_KEYS, _RETINA_ROWS)
                  if k in keys]
        return d

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas,
             gt_bboxes_ignore=None):
        """retina_gfl_head.py:157-229: loss_cls, loss_bbox."""
        self._check_loss_cfg()
        return self._run_block(self._hp(), cls_scores, bbox_preds, gt_bboxes,
                               gt_labels, img_metas, cls_scores, bbox_preds,
                               ('loss_cls', 'loss_bbox'))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None,
                      gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*self(x), gt_bboxes, gt_labels, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None,
                   rescale=False, with_nms=True):
        """anchor_head.py:497-589 + retina_gfl_head.py:301-412: sigmoid scores,
        Integral * stride, per-level top-nms_pre over all (cell, base anchor)
        rows, decode about the cell centre, multiclass_nms."""
        return GFLHead._get_bboxes(self, cls_scores, bbox_preds, img_metas,
                                   cfg, rescale, with_nms, prob=False,
                                   num_base=self.num_anchors)


@HEADS.register_module()
class LDRetinaHead(RetinaGFLHead):
    """ld_retina.py:13-636: localization distillation on the RetinaGFL head --
    LD over the 68 corner logits of every positive anchor weighted by its max
    class score, 0.03 x the same on the valuable localisation region of the
    background anchors, KD on the positives' class logits.  The detector calls
    it with output_feature=False: forward_train(x, out_teacher, img_metas,
    ...)."""

    def __init__(self, num_classes, in_channels,
                 loss_ld=dict(type='LocalizationDistillationLoss',
                              loss_weight=0.25, T=10),
                 loss_kd=None, **kwargs):
        super().__init__(num_classes, in_channels, **kwargs)
        self.loss_ld = build_loss(loss_ld)
        self.loss_kd = build_loss(loss_kd)

    def forward_train(self, x, out_teacher, img_metas, gt_bboxes,
                      gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """ld_retina.py:139-185."""
        if gt_labels is None:
            raise NotImplementedError('LDRetinaHead needs gt_labels')
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*self(x), gt_bboxes, gt_labels, out_teacher,
                         img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, out_teacher,
             img_metas, gt_bboxes_ignore=None):
        """ld_retina.py:187-254 -> the five keys of RETINA_LOSS_KEYS."""
        self._check_loss_cfg()
        soft_labels, soft_targets = out_teacher[0], out_teacher[1]
        # loss_ld_vlr = 0.03 * loss_ld(..., avg_factor=4) (ld_retina.py:109-110);
        # the block's VLR term is lw_ld_vlr * sum / 16
        hp = self._hp(lw_ld=self.loss_ld.loss_weight, T_ld=self.loss_ld.T,
                      lw_ld_vlr=0.03 * 4.0 * self.loss_ld.loss_weight,
                      T_ld_vlr=self.loss_ld.T,
                      lw_kd=self.loss_kd.loss_weight, T_kd=self.loss_kd.T)
        return self._run_block(hp, cls_scores, bbox_preds, gt_bboxes,
                               gt_labels, img_metas, soft_labels, soft_targets,
                               RETINA_LOSS_KEYS)


class _Marker(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential's indices (and with
    them the state_dict keys ``reg_conf.0.*`` / ``reg_conf.2.*``) identical to
    the reference's [Conv2d, ReLU, Conv2d, Sigmoid]; the arithmetic of all four
    stages is the fused quality kernel."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what


@HEADS.register_module()
class GFocalHead(GFLHead):
    """GFLv2 head (gfocal_head.py:14-217): GFLHead's towers + the
    distribution-guided quality estimator ``reg_conf``; cls_out_channels is
    num_classes + 1 because its QFL runs with use_sigmoid=False
    (anchor_head.py:68-71).  forward returns (cls_scores, bbox_preds,
    cls_feats) like the reference."""

    def __init__(self, num_classes, in_channels, stacked_convs=4,
                 conv_cfg=None,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
                 reg_max=16, reg_topk=4, reg_channels=64, add_mean=True,
                 **kwargs):
        if (reg_topk, reg_channels, bool(add_mean), reg_max) != (4, 64, True,
                                                                 16):
            raise NotImplementedError(
                'GFocalHead: the fused quality kernel is compiled for '
                'reg_topk=4, reg_channels=64, add_mean=True, reg_max=16 '
                '(the values of configs/gfl/gflv2_*.py and configs/ldv2)')
        self.reg_topk, self.reg_channels, self.add_mean = (reg_topk,
                                                           reg_channels,
                                                           add_mean)
        self.total_dim = reg_topk + (1 if add_mean else 0)
        super().__init__(num_classes, in_channels, stacked_convs=stacked_convs,
                         conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                         loss_dfl=loss_dfl, reg_max=reg_max, **kwargs)

    def _init_layers(self):
        """gfocal_head.py:102-144."""
        super()._init_layers()
        self.reg_conf = nn.Sequential(
            Conv2d(4 * self.total_dim, self.reg_channels, 1), _Marker('ReLU'),
            Conv2d(self.reg_channels, 1, 1), _Marker('Sigmoid'))

    def init_weights(self):
        """gfocal_head.py:146-158."""
        super().init_weights()
        for m in self.reg_conf:
            if isinstance(m, Conv2d):
                normal_init(m, std=0.01)

    def forward(self, feats):
        """gfocal_head.py:160-217, all levels per launch: towers, predictors,
        per-level Scale, then the fused quality kernel."""
        assert len(feats) == len(self.scales)
        x3, levels = self._pack(feats)
        cls_feat = reg_feat = x3
        cls_feat = _tower(self.cls_convs, cls_feat, levels)
        reg_feat = _tower(self.reg_convs, reg_feat, levels)
        cls3, _ = self.gfl_cls.forward3(cls_feat, levels)
        reg3, _ = self.gfl_reg.forward3(reg_feat, levels)
        scales = torch.stack([s.scale for s in self.scales])
        reg3 = Y.scale_levels(reg3, scales, levels)
        c0, c2 = self.reg_conf[0], self.reg_conf[2]
        score3, _ = Y.QualityFn.apply(reg3, cls3, c0.weight, c0.bias,
                                      c2.weight, c2.bias)
        return (Y.split_levels(score3, levels), Y.split_levels(reg3, levels),
                Y.split_levels(cls3, levels))

    def _hp(self, **over):
        over.setdefault('cls_channels', self.cls_out_channels)
        over.setdefault('flags', L.LD_LOSS_PROB_CLS)
        return super()._hp(**over)

    def _check_loss_cfg(self):
        super()._check_loss_cfg()
        if getattr(self.loss_cls, 'use_sigmoid', True):
            raise NotImplementedError(
                'GFocalHead multiplies sigmoid(cls_feat) by the quality score '
                'itself: its QualityFocalLoss must have use_sigmoid=False '
                '(configs/gfl/gflv2_*.py)')

    def loss(self, cls_scores, bbox_preds, cls_feat, gt_bboxes, gt_labels,
             img_metas, gt_bboxes_ignore=None):
        """gfocal_head.py:230-352 (plain GFLv2: QFL on probabilities + GIoU +
        DFL); the distillation terms of the fused block run with weight 0 on
        the student's own detached maps."""
        self._check_loss_cfg()
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        device = cls_scores[0].device
        hp = self._hp()
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, hp, device)
        det = [c.detach() for c in cls_feat]
        hp.feat_channels = cls_feat[0].shape[1]
        teacher = (det, [b.detach() for b in bbox_preds], det, det)
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream, *cls_scores,
                                        *bbox_preds, *det, *cls_feat)
        return self._loss_dict(table, ('loss_cls', 'loss_bbox', 'loss_dfl'))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None,
                      gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        outs = self(x)
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*outs, gt_bboxes, gt_labels, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    def get_bboxes(self, cls_scores, bbox_preds, cls_feat, img_metas, cfg=None,
                   rescale=False, with_nms=True):
        """gfocal_head.py:317-596: GFLHead's pipeline on the head's own
        probabilities (cls_score = sigmoid(cls) * quality, no second sigmoid)
        over all cls_out_channels = num_classes + 1 score channels -- the
        reference's background column is an ordinary class here, label 80
        included.  ``cls_feat`` (the third head output) is unused, as in the
        reference."""
        return self._get_bboxes(cls_scores, bbox_preds, img_metas, cfg,
                                rescale, with_nms, prob=True)


@HEADS.register_module()
class LDv2Head(GFocalHead):
    """ld_gflv2.py:44-644: LDHead's distillation terms on a GFocalHead.
    Differences from LDHead that the fused block is told through its hp flags
    (ld_gflv2.py:200,243,326): weight_targets = max_c cls_score with no
    sigmoid, QFL on probabilities over 81 channels, KD on the raw cls_feat of
    student and teacher (``soft_teacher`` = (cls_score, bbox_pred, cls_feat),
    of which the first is ignored)."""
    _wants_packed_feats = True

    def __init__(self, num_classes, in_channels,
                 loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                              loss_weight=0.25, T=10),
                 loss_ld_vlr=dict(type='KnowledgeDistillationKLDivLoss',
                                  loss_weight=0.25, T=10),
                 loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                              loss_weight=10, T=2),
                 loss_im=dict(type='IMLoss', loss_weight=0),
                 imitation_method='gibox', **kwargs):
        super().__init__(num_classes, in_channels, **kwargs)
        assert imitation_method in ['gibox', 'finegrained', 'fitnet',
                                    'decouple']
        self.imitation_method = imitation_method
        self.loss_im = build_loss(loss_im)
        self.loss_ld = build_loss(loss_ld)
        self.loss_ld_vlr = build_loss(loss_ld_vlr)
        self.loss_kd = build_loss(loss_kd)
        self.iou_calculator = build_iou_calculator(dict(type='BboxOverlaps2D'))

    def forward_train(self, x, out_teacher, teacher_x, img_metas, gt_bboxes,
                      gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """ld_gflv2.py:74-114."""
        outs = self(x)
        if gt_labels is None:
            raise NotImplementedError('LDv2Head needs gt_labels')
        if proposal_cfg is not None:
            raise NotImplementedError('proposal_cfg')
        return self.loss(*outs, gt_bboxes, gt_labels, out_teacher,
                         self._loss_feats(x), teacher_x, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    def loss(self, cls_scores, bbox_preds, cls_feat, gt_bboxes, gt_labels,
             soft_teacher, x, teacher_x, img_metas, gt_bboxes_ignore=None):
        """ld_gflv2.py:286-380 -> dict of 8 lists of per-level scalars."""
        self._check_loss_cfg()
        lw_im = float(self.loss_im.loss_weight)
        im_flags = _imitation_flags(self.imitation_method, lw_im)
        if x[0].shape[1] != 256:
            raise ValueError('LDv2Head hard-codes 256 feature channels '
                             '(ld_gflv2.py:155-156)')
        _, soft_target, soft_label = soft_teacher  # ld_gflv2.py:326
        sizes = [tuple(int(v) for v in f.shape[-2:]) for f in cls_scores]
        assert len(sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        hp = self._hp(lw_ld=self.loss_ld.loss_weight, T_ld=self.loss_ld.T,
                      lw_ld_vlr=self.loss_ld_vlr.loss_weight,
                      T_ld_vlr=self.loss_ld_vlr.T,
                      lw_kd=self.loss_kd.loss_weight, T_kd=self.loss_kd.T,
                      lw_im=lw_im)
        hp.flags |= im_flags
        targets = self.get_targets_batched(sizes, img_metas, gt_bboxes,
                                           gt_labels, hp, device)
        t_kd = [t.detach() for t in soft_label]
        teacher = (t_kd, [t.detach() for t in soft_target],
                   [t.detach() for t in teacher_x], t_kd)
        if self.imitation_method == 'gibox' and lw_im != 0.0:
            # ld_gflv2.py:619-644: raw teacher cls_feat vs the student's
            # probabilities, no sigmoids
            targets = LB.gi_region(hp, targets,
                                   [c.detach() for c in cls_scores],
                                   [b.detach() for b in bbox_preds], t_kd,
                                   teacher[1])
        table, _ = LB.LDLossBlock.apply(hp, targets, teacher,
                                        self._norm_reducer(),
                                        self.unit_upstream, *cls_scores,
                                        *bbox_preds, *x, *cls_feat)
        self.last_targets = targets
        return self._loss_dict(table)
