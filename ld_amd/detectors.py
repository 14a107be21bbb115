"""Detector wrappers with mmdet's registry names and methods (reference:
mmdet/models/detectors/{base,single_stage,gfl,kd_one_stage}.py)."""
import os
import warnings
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from .config import Config, ConfigDict
from .heads import LazyScalars
from . import layers as Y
from .layers import side_stream as _side_stream
from .registry import (DETECTORS, build_backbone, build_detector, build_head,
                       build_neck)


_INDEX_CACHE = {}


def _device_index(values, device):
    """A small constant int64 index tensor on ``device``, built once."""
    key = (str(device), values)
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = torch.tensor(values, dtype=torch.long, device=device)
        _INDEX_CACHE[key] = t
    return t


def _allow_missing_ckpt():
    """Opt-in (bench / smoke / offline config tests) for running with a
    checkpoint that cannot be found; the default is the reference's: raise."""
    return os.environ.get('LD_ALLOW_MISSING_CKPT', '0') == '1'


@DETECTORS.register_module()
class SingleStageDetector(nn.Module):
    """single_stage.py:10-57 + base.py:16-268 (train path)."""

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None):
        super().__init__()
        self.fp16_enabled = False
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=ConfigDict.wrap(train_cfg) if train_cfg
                         else train_cfg)
        bbox_head.update(test_cfg=ConfigDict.wrap(test_cfg) if test_cfg
                         else test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    def init_weights(self, pretrained=None):
        """single_stage.py:35-50."""
        if isinstance(pretrained, str):
            from .checkpoint import resolve_checkpoint_path
            try:
                pretrained = resolve_checkpoint_path(pretrained)
            except FileNotFoundError as e:
                # mmcv's load_checkpoint raises here, and so do we: a randomly
                # initialised backbone trains to finite, healthy-looking, wrong
                # numbers.  Synthetic-throughput runs opt in explicitly.
                if not _allow_missing_ckpt():
                    raise
                warnings.warn(f'{e}; LD_ALLOW_MISSING_CKPT=1: initialising '
                              'the backbone randomly instead')
                pretrained = None
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        self.bbox_head.init_weights()

    def extract_feat(self, img):
        if torch.is_grad_enabled():
            # bf16 mode: the trainable trunk hands its activations from block to
            # block (and to the neck's lateral convs) as bf16 C8 images only
            # (layers.trunk_c8_scope; a no-op in fp32 mode)
            with Y.trunk_c8_scope():
                x = self.backbone(img)
        else:
            x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels,
                      gt_bboxes_ignore=None):
        x = self.extract_feat(img)
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels,
                                            gt_bboxes_ignore)

    def simple_test(self, img, img_metas, rescale=False):
        """single_stage.py:98-129: features -> head -> get_bboxes (one device
        call, ld_get_bboxes) -> per-class numpy arrays."""
        from .core import bbox2result
        x = self.extract_feat(img)
        outs = self.bbox_head(x)
        bbox_list = self.bbox_head.get_bboxes(*outs, img_metas,
                                              rescale=rescale)
        return [bbox2result(b, l, self.bbox_head.num_classes)
                for b, l in bbox_list]

    def aug_test(self, imgs, img_metas, rescale=False):
        raise NotImplementedError('test-time augmentation (single_stage.py:'
                                  '131-160) is not on the SURVEY.md section 8 '
                                  'scope')

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:120-167: outer lists = test-time augmentations."""
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) '
                             f'!= num of image meta ({len(img_metas)})')
        for img, img_meta in zip(imgs, img_metas):
            for m in img_meta:
                m['batch_input_shape'] = tuple(img.size()[-2:])
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """base.py:169-183."""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        with torch.no_grad():
            return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """base.py:185-218.  Same keys and values; computed on the device in
        O(1) launches, with ONE all-reduce for all log values and a lazy D2H
        copy (the reference does one all-reduce + .item() per key)."""
        table = getattr(losses, 'table', None)
        if table is not None:
            names = list(losses.keys())
            rows = list(losses.rows)
            # row selection by slicing / a cached device index: indexing with a
            # Python list would build the index on the host and copy it over
            # synchronously (a host sync per step, and illegal under hipGraph
            # capture)
            if rows == list(range(rows[0], rows[0] + len(rows))):
                key_sums = table[rows[0]:rows[0] + len(rows)].sum(1)
            else:
                key_sums = table.index_select(
                    0, _device_index(tuple(rows), table.device)).sum(1)
        else:
            names, vals = [], []
            for name, value in losses.items():
                if isinstance(value, torch.Tensor):
                    vals.append(value.mean())
                elif isinstance(value, list):
                    vals.append(sum(v.mean() for v in value))
                else:
                    raise TypeError(f'{name} is not a tensor or list of tensors')
                names.append(name)
            key_sums = torch.stack(vals)
        sel = [i for i, k in enumerate(names) if 'loss' in k]
        if len(sel) == len(names):
            loss = key_sums.sum()
        else:
            loss = key_sums.index_select(
                0, _device_index(tuple(sel), key_sums.device)).sum()
        logged = torch.cat([key_sums.detach(), loss.detach().reshape(1)])
        from .train import _diag_skip, collectives_on
        if collectives_on() and not _diag_skip('logs'):
            logged = logged.clone()
            dist.all_reduce(logged.div_(dist.get_world_size()))
        log_vars = LazyScalars(names + ['loss'], logged)
        return loss, log_vars

    def train_step(self, data, optimizer):
        """base.py:220-253."""
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars,
                    num_samples=len(data['img_metas']))


@DETECTORS.register_module()
class GFL(SingleStageDetector):
    """gfl.py:6-16."""


@DETECTORS.register_module()
class ATSS(SingleStageDetector):
    """atss.py:6-16 (the detector type of configs/gfl/atss_gfl_*.py)."""


@DETECTORS.register_module()
class FCOS(SingleStageDetector):
    """fcos.py:6-17 (the detector type of configs/gfl/fcos_gfl_*.py)."""


@DETECTORS.register_module()
class RetinaNet(SingleStageDetector):
    """retinanet.py:6-17 (the detector type of configs/gfl/retinagfl_*.py)."""


@DETECTORS.register_module()
class KnowledgeDistillationSingleStageDetector(SingleStageDetector):
    """kd_one_stage.py:12-108: student + frozen teacher (hidden from
    ``parameters()`` / ``state_dict()``), dual forward, LD loss."""

    def __init__(self, backbone, neck, bbox_head, teacher_config,
                 output_feature=False, teacher_ckpt=None, eval_teacher=True,
                 train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg,
                         pretrained)
        self.eval_teacher = eval_teacher
        self.output_feature = output_feature
        if isinstance(teacher_config, str):
            teacher_config = Config.fromfile(teacher_config)
        tcfg = dict(teacher_config['model'])
        self.teacher_model = build_detector(tcfg)
        if eval_teacher:
            # the teacher only ever runs under no_grad (kd_one_stage.py:70-72)
            # and no optimizer owns it: tag its parameters static so the conv
            # weight images and folded-BN coefficients are computed once
            # instead of every step (requires_grad is left as the reference
            # has it)
            for p in self.teacher_model.parameters():
                p._ld_static = True
            # bf16 mode: its trunk activations exist only as bf16 C8 images
            # (resnet.ResNet._c8_only; the fp32 modes are unaffected)
            self.teacher_model.backbone.c8_activations = True
        if teacher_ckpt is not None:
            from .checkpoint import load_checkpoint
            try:
                load_checkpoint(self.teacher_model, teacher_ckpt,
                                map_location='cpu')
            except FileNotFoundError as e:
                # kd_one_stage.py:42-44 lets mmcv raise; distilling from a
                # random teacher gives finite LD/KD/IM losses that mean nothing
                if not _allow_missing_ckpt():
                    raise
                warnings.warn(f'LD_ALLOW_MISSING_CKPT=1: teacher_ckpt not '
                              f'loaded ({e}); the teacher is RANDOM')
        # run the (independent) teacher forward on its own HIP stream so it
        # overlaps the student's: under-filled launches of one net are
        # back-filled by the other
        self.teacher_stream = None
        self.use_teacher_stream = os.environ.get('LD_TEACHER_STREAM',
                                                 '1') == '1'

    def _teacher_forward(self, img):
        if self._replay_enabled(img):
            return self._teacher_replay(img)
        with torch.no_grad():
            teacher_x = self.teacher_model.extract_feat(img)
            out_teacher = self.teacher_model.bbox_head(teacher_x)
        return teacher_x, out_teacher

    # ---- the frozen teacher as launch lists (round 5) -------------------------
    # kd_one_stage.py:70-72 runs the teacher under no_grad in eval mode: the same
    # ~150 launches on the same shapes every step, a quarter of the step's
    # launches and ~3.5 ms of Python / ctypes per step.  After one ordinary
    # forward per (image shape, precision) -- which fills the weight-image / BN /
    # workspace caches -- the next TEACHER_SLOTS forwards are RECORDED
    # (ld_record_begin .. ld_record_end, include/ld_hip.h "launch lists"), each
    # into its own set of buffers; from then on a teacher forward is one copy of
    # the batch into a slot's input buffer + ONE C call that re-issues that slot's
    # launches.  Slots rotate so that a result is not overwritten while the step
    # it was computed for (up to two steps of look-ahead, prefetch_teacher) still
    # reads it.  The kernels, operands and results are those of the ordinary
    # forward, bit for bit (tests/test_gpu_teacher_replay.py).
    TEACHER_SLOTS = 3

    def _teacher_fingerprint(self):
        ts = getattr(self, '_teacher_tensors', None)
        if ts is None:
            ts = list(self.teacher_model.parameters()) + \
                list(self.teacher_model.buffers())
            object.__setattr__(self, '_teacher_tensors', ts)
        # every tensor's storage AND version: a `p.data = ...` reassignment or a
        # re-materialised middle parameter changes a data_ptr but no _version
        v, h = 0, 0
        for i, t in enumerate(ts):
            v += t._version
            h = (h * 1000003 + t.data_ptr() + i) & 0xFFFFFFFFFFFFFFFF
        return (v, h, len(ts))

    def _replay_enabled(self, img):
        from . import layers as Y
        return (self.eval_teacher and img.is_cuda and img.is_contiguous() and
                os.environ.get('LD_TEACHER_REPLAY', '1') == '1' and
                not self.teacher_model.training and
                Y.KernelProfile.active is None and not Y._AUTOTUNE[0])

    def reset_teacher_replay(self):
        """Drop the recorded teacher launch lists (they are re-recorded on
        demand): after anything the fingerprint cannot see."""
        from . import lib as L
        plans = getattr(self, '_tplans', None) or {}
        if plans and L.lib_available():
            lib = L.get_lib()
            for pl in plans.values():
                for sl in pl['slots']:
                    lib.ld_record_free(sl['handle'])
        object.__setattr__(self, '_tplans', {})

    def _teacher_replay(self, img):
        from . import layers as Y
        from . import lib as L
        plans = getattr(self, '_tplans', None)
        if plans is None:
            plans = {}
            object.__setattr__(self, '_tplans', plans)
        fp = self._teacher_fingerprint()
        key = (tuple(img.shape), str(img.device), Y.get_precision(), Y._C8[0],
               Y._FUSED_BLOCK[0])
        pl = plans.get(key)
        if pl is not None and pl['fp'] != fp:  # the teacher's weights changed
            self.reset_teacher_replay()
            plans, pl = self._tplans, None
        if pl is None:
            pl = plans[key] = dict(fp=fp, warm=False, slots=[], next=0)
        capturing = torch.cuda.is_current_stream_capturing()
        if not pl['warm'] or (capturing and
                              len(pl['slots']) < self.TEACHER_SLOTS):
            # the ordinary forward: fills the caches a recording must not contain
            # (weight transforms, BN folding, workspaces); also what a hipGraph
            # capture takes until the slots exist (recording allocates)
            with torch.no_grad():
                teacher_x = self.teacher_model.extract_feat(img)
                out_teacher = self.teacher_model.bbox_head(teacher_x)
            pl['warm'] = True
            return teacher_x, out_teacher
        lib = L.get_lib()
        if len(pl['slots']) < self.TEACHER_SLOTS:
            sl = dict(img=torch.empty_like(img), keep=[])
            sl['img'].copy_(img)
            from . import lossblock as LB
            L._KEEP[0] = sl['keep']
            # scratch buffers private to this list (lossblock.workspace)
            LB._WS_SCOPE[0] = ('teacher-list', id(self), key, len(pl['slots']))
            L.check(lib.ld_record_begin(), 'ld_record_begin')
            try:
                with torch.no_grad():
                    teacher_x = self.teacher_model.extract_feat(sl['img'])
                    out_teacher = self.teacher_model.bbox_head(teacher_x)
            except BaseException:
                lib.ld_record_abort()
                raise
            finally:
                L._KEEP[0] = None
                LB._WS_SCOPE[0] = None
            sl['handle'] = lib.ld_record_end()
            if sl['handle'] <= 0:
                raise L.LdError(f'ld_record_end failed ({sl["handle"]})')
            sl['out'] = (teacher_x, out_teacher)
            sl['launches'] = lib.ld_record_count(sl['handle'])
            pl['slots'].append(sl)
            return teacher_x, out_teacher
        sl = pl['slots'][pl['next']]
        pl['next'] = (pl['next'] + 1) % len(pl['slots'])
        if img.data_ptr() != sl['img'].data_ptr():
            Y.copy_into(sl['img'], img)  # a kernel node if this step is captured
        L.check(lib.ld_record_replay(sl['handle'], L.stream_ptr(img.device)),
                'ld_record_replay')
        self.teacher_replays = getattr(self, 'teacher_replays', 0) + 1
        return sl['out']

    @staticmethod
    def _same_batch(entry, img):
        """A prefetched entry belongs to ``img`` iff it was computed from that
        very tensor OBJECT at its current version.  (Round 2 compared
        data_ptr / _version / shape: a freed batch whose storage the caching
        allocator hands to the next one collides on all three.)  The queue
        keeps the tensor alive, so identity is meaningful."""
        return entry['img'] is img and entry['version'] == img._version

    def prefetch_teacher(self, img):
        """Software pipelining across steps.  The frozen teacher depends on the
        batch only (kd_one_stage.py:70-72: no_grad, eval), so its forward for
        the NEXT batch can be enqueued on the teacher stream now and run under
        this step's student forward AND backward, filling the CUs the
        under-filled student launches (50x84 / 25x42 stages, wgrad at 2 waves
        per SIMD) leave idle.  The forward_train call that later receives this
        same ``img`` tensor (same storage, same version) consumes the result;
        any other image falls back to the in-step teacher forward.  Numerically
        identical: the same kernels on the same inputs."""
        if not (self.use_teacher_stream and img.is_cuda and self.eval_teacher):
            return False
        if self.teacher_stream is None:
            self.teacher_stream = _side_stream(img.device, 'teacher')
        main = torch.cuda.current_stream(img.device)
        side = self.teacher_stream
        side.wait_stream(main)  # the batch (and all earlier work) is ready
        with torch.cuda.stream(side):
            teacher_x, out_teacher = self._teacher_forward(img)
            done = torch.cuda.Event()
            done.record(side)
        # FIFO: the prefetch for step i + 1 is enqueued before step i consumes
        # its own
        if not hasattr(self, '_prefetched') or self._prefetched is None:
            self._prefetched = []
        self._prefetched.append(dict(img=img, version=img._version,
                                     teacher_x=teacher_x,
                                     out_teacher=out_teacher, done=done))
        # a caller that announces batches it never trains on must not pin
        # their teacher outputs forever: two steps of look-ahead at most
        while len(self._prefetched) > 2:
            self._prefetched.pop(0)
            self.prefetch_dropped = getattr(self, 'prefetch_dropped', 0) + 1
        return True

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels,
                      gt_bboxes_ignore=None):
        """kd_one_stage.py:46-81."""
        if not self.output_feature and \
                type(self.bbox_head).__name__ in ('LDHead', 'LDv2Head'):
            raise NotImplementedError(
                'output_feature=False: the reference passes one argument too '
                'few to LDHead.forward_train on that branch '
                '(kd_one_stage.py:74-76 vs ld_head.py:73-82)')
        forced = getattr(self, '_forced_teacher', None)
        if forced is not None:
            # train.PipelinedGraphedStep: the teacher outputs of THIS batch were
            # computed by the previous graph replay into static buffers
            teacher_x, out_teacher = forced
            x = self.extract_feat(img)
            return self._head_train(x, out_teacher, teacher_x, img_metas,
                                    gt_bboxes, gt_labels, gt_bboxes_ignore)
        side = None
        queue = getattr(self, '_prefetched', None) or []
        pre = None
        # SGDTrainer.step enqueues the prefetch of batch i + 1 BEFORE batch i
        # gets here, so at step 0 the queue holds [pre(1)] and no entry matches:
        # that is not staleness (round 2 cleared the queue there, and then
        # never hit again).  Take the first entry made from THIS tensor; entries
        # ahead of it were announced but skipped and are dropped; on no match
        # the queue is left alone (prefetch_teacher bounds its length).
        for i, e in enumerate(queue):
            if self._same_batch(e, img):
                pre = e
                self.prefetch_dropped = getattr(self, 'prefetch_dropped', 0) + i
                del queue[:i + 1]
                break
        if pre is not None:
            self.prefetch_hits = getattr(self, 'prefetch_hits', 0) + 1
            main = torch.cuda.current_stream(img.device)
            teacher_x, out_teacher = pre['teacher_x'], pre['out_teacher']
            x = self.extract_feat(img)
            # wait for THAT forward only: the stream may already hold the
            # prefetch of the following batch
            main.wait_event(pre['done'])
            for t in list(teacher_x) + [t for lvl in out_teacher for t in lvl]:
                t.record_stream(main)
            return self._head_train(x, out_teacher, teacher_x, img_metas,
                                    gt_bboxes, gt_labels, gt_bboxes_ignore)
        if self.use_teacher_stream and img.is_cuda:
            if self.teacher_stream is None:
                self.teacher_stream = _side_stream(img.device, 'teacher')
            main = torch.cuda.current_stream(img.device)
            side = self.teacher_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                teacher_x, out_teacher = self._teacher_forward(img)
        x = self.extract_feat(img)
        if side is not None:
            main.wait_stream(side)
            # GFLHead returns (cls, reg), GFocalHead (cls_score, reg, cls_feat)
            for t in list(teacher_x) + [t for lvl in out_teacher for t in lvl]:
                t.record_stream(main)
        else:
            teacher_x, out_teacher = self._teacher_forward(img)
        return self._head_train(x, out_teacher, teacher_x, img_metas,
                                gt_bboxes, gt_labels, gt_bboxes_ignore)

    def _head_train(self, x, out_teacher, teacher_x, img_metas, gt_bboxes,
                    gt_labels, gt_bboxes_ignore):
        """kd_one_stage.py:73-80: the feature-imitation heads take teacher_x,
        the others (LDATSSHead, ...) do not."""
        if not self.output_feature:
            return self.bbox_head.forward_train(x, out_teacher, img_metas,
                                                gt_bboxes, gt_labels,
                                                gt_bboxes_ignore)
        return self.bbox_head.forward_train(x, out_teacher, teacher_x,
                                            img_metas, gt_bboxes, gt_labels,
                                            gt_bboxes_ignore)

    def cuda(self, device=None):
        self.teacher_model.cuda(device=device)
        return super().cuda(device=device)

    def to(self, *args, **kwargs):
        self.teacher_model.to(*args, **kwargs)
        return super().to(*args, **kwargs)

    def train(self, mode=True):
        if self.eval_teacher:
            self.teacher_model.train(False)
        else:
            self.teacher_model.train(mode)
        return super().train(mode)

    def __setattr__(self, name, value):
        """kd_one_stage.py:97-108: keep the teacher a plain attribute."""
        if name == 'teacher_model':
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)
