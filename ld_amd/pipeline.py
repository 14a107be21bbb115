"""Device input pipeline + distributed group sampler.

The reference feeds the train step from CPU dataloader workers
(data.workers_per_gpu=2, configs/_base_/datasets/coco_detection.py) running

    Resize(img_scale=(1333, 800), keep_ratio=True) -> RandomFlip(0.5) ->
    Normalize(mean, std, to_rgb=True) -> Pad(size_divisor=32) ->
    DefaultFormatBundle -> Collect -> mmcv.parallel.collate

(configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py:66-77;
mmdet/datasets/pipelines/transforms.py:31-300 Resize, :303-460 RandomFlip,
:463-540 Pad, :543-590 Normalize).  At the MI355X step rate (50-90 images/s
per GPU) two CPU workers per GPU cannot keep up, so here everything after the
JPEG decode runs on the device: the decoded uint8 HWC images are copied to HBM
as they are (3 bytes / pixel over PCIe instead of 12) and ONE launch of
``ld_preprocess_batch`` (csrc/pipeline.hip) writes the padded fp32 NCHW batch.
The per-image bookkeeping the reference keeps in ``img_metas`` and the GT-box
transforms are host integer / float arithmetic restated below with the same
rounding; GT boxes are tiny and go to the device with the same async copy.

``DistributedGroupSampler`` restates mmdet/datasets/samplers/group_sampler.py:
51-147 (the sampler build_dataloader picks for dist=True) index for index; it
is pinned against the reference's own sampler in tests/test_pipeline.py.

There is no CPU fallback for the image arithmetic: DevicePipeline refuses to
run without the HIP library and a device.
"""
import ctypes as C
import math

import numpy as np
import torch


# ---------------------------------------------------------------------------
# host arithmetic (mmcv.imrescale's size rule, box transforms, img_metas)
# ---------------------------------------------------------------------------
def rescale_size(old_size, scale):
    """mmcv.image.geometric.rescale_size (mmcv 1.2.x, the version the
    reference pins in requirements): ``old_size`` (w, h); ``scale`` a
    (long_edge, short_edge) pair in any order, or a float factor."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        factor = scale
    elif isinstance(scale, tuple):
        max_long_edge = max(scale)
        max_short_edge = min(scale)
        factor = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    else:
        raise TypeError(
            f'Scale must be a number or tuple of int, but got {type(scale)}')
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def sample_scale(img_scale, multiscale_mode='range', ratio_range=None,
                 rng=np.random):
    """Resize._random_scale (transforms.py:170-200): the draw order and the
    draws themselves are the reference's, so that the same numpy seed picks the
    same scale."""
    scales = img_scale if isinstance(img_scale, list) else [img_scale]
    if ratio_range is not None:
        lo, hi = ratio_range
        ratio = rng.random_sample() * (hi - lo) + lo
        return int(scales[0][0] * ratio), int(scales[0][1] * ratio)
    if len(scales) == 1:
        return tuple(scales[0])
    if multiscale_mode == 'range':
        longs = [max(s) for s in scales]
        shorts = [min(s) for s in scales]
        long_edge = rng.randint(min(longs), max(longs) + 1)
        short_edge = rng.randint(min(shorts), max(shorts) + 1)
        return long_edge, short_edge
    if multiscale_mode == 'value':
        return tuple(scales[rng.randint(len(scales))])
    raise NotImplementedError(multiscale_mode)


def resize_bboxes(bboxes, scale_factor, img_shape, clip=True):
    """Resize._resize_bboxes (transforms.py:233-241); fp32 like the reference
    (gt boxes are float32, scale_factor is a float32 4-vector)."""
    b = np.asarray(bboxes, np.float32).reshape(-1, 4) * scale_factor
    if clip:
        b[:, 0::2] = np.clip(b[:, 0::2], 0, img_shape[1])
        b[:, 1::2] = np.clip(b[:, 1::2], 0, img_shape[0])
    return b


def flip_bboxes(bboxes, img_shape):
    """RandomFlip.bbox_flip, direction='horizontal' (transforms.py:384-401)."""
    b = np.asarray(bboxes, np.float32)
    out = b.copy()
    w = img_shape[1]
    out[..., 0::4] = w - b[..., 2::4]
    out[..., 2::4] = w - b[..., 0::4]
    return out


def _ceil_to(v, d):
    return int(math.ceil(v / d)) * d


class DevicePipeline:
    """The reference's train_pipeline from Resize to collate, on the device.

    ``__call__(images, gt_bboxes, gt_labels)`` takes per-image decoded uint8 HWC
    arrays (BGR, what LoadImageFromFile produces) and float32 (k, 4) / int64
    (k,) annotations; returns the dict the detector's ``forward_train`` takes:
    ``img`` (N, 3, Hpad, Wpad) fp32 on the device, ``img_metas`` (the keys of
    Collect's default meta_keys that the heads read), ``gt_bboxes`` /
    ``gt_labels`` lists of device tensors.
    """

    def __init__(self, img_scale=(1333, 800), keep_ratio=True, flip_ratio=0.5,
                 mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375),
                 to_rgb=True, size_divisor=32, multiscale_mode='range',
                 ratio_range=None, bbox_clip_border=True, device=None):
        if not keep_ratio:
            raise NotImplementedError(
                'keep_ratio=False is not used by any configs/ld or configs/ldv2 '
                'train pipeline')
        self.img_scale = img_scale
        self.multiscale_mode = multiscale_mode
        self.ratio_range = ratio_range
        self.flip_ratio = float(flip_ratio or 0.0)
        self.mean = np.asarray(mean, np.float32)
        self.std = np.asarray(std, np.float32)
        # mmcv.imnormalize: stdinv = 1 / np.float64(std.reshape(1, -1)), then
        # cv2.multiply on the float32 image
        self.std_inv = (1.0 / self.std.astype(np.float64)).astype(np.float32)
        self.to_rgb = bool(to_rgb)
        self.size_divisor = size_divisor
        self.bbox_clip_border = bbox_clip_border
        self.device = torch.device(device if device is not None else 'cuda')

    @classmethod
    def from_cfg(cls, train_pipeline, device=None):
        """Build from a reference config's ``train_pipeline`` list (the dicts
        with type Resize / RandomFlip / Normalize / Pad); other entries are the
        loading / formatting steps this class subsumes."""
        kw = {}
        for step in train_pipeline:
            t = step['type']
            if t == 'Resize':
                kw['img_scale'] = step['img_scale']
                kw['keep_ratio'] = step.get('keep_ratio', True)
                kw['multiscale_mode'] = step.get('multiscale_mode', 'range')
                kw['ratio_range'] = step.get('ratio_range')
            elif t == 'RandomFlip':
                kw['flip_ratio'] = step.get('flip_ratio')
            elif t == 'Normalize':
                kw['mean'], kw['std'] = step['mean'], step['std']
                kw['to_rgb'] = step.get('to_rgb', True)
            elif t == 'Pad':
                kw['size_divisor'] = step.get('size_divisor')
            elif t not in ('LoadImageFromFile', 'LoadAnnotations',
                           'DefaultFormatBundle', 'Collect'):
                raise NotImplementedError(f'pipeline step {t}')
        return cls(device=device, **kw)

    # -- per-image host decisions -------------------------------------------
    def plan(self, shapes, rng=np.random):
        """For each (h, w): the scale, the resized size, scale_factor and the
        flip decision, drawing from ``rng`` in the reference's order (scale
        draws of Resize, then RandomFlip's one np.random.choice)."""
        plans = []
        for (h, w) in shapes:
            scale = sample_scale(self.img_scale, self.multiscale_mode,
                                 self.ratio_range, rng)
            new_w, new_h = rescale_size((w, h), tuple(scale))
            flip = False
            if self.flip_ratio > 0:
                # np.random.choice(['horizontal', None], p=[r, 1 - r])
                flip = int(rng.choice(2, p=[self.flip_ratio,
                                            1 - self.flip_ratio])) == 0
            sf = np.array([new_w / w, new_h / h, new_w / w, new_h / h],
                          np.float32)
            plans.append(dict(ori_shape=(h, w, 3), img_shape=(new_h, new_w, 3),
                              scale_factor=sf, flip=flip, scale=tuple(scale)))
        return plans

    def transform_boxes(self, bboxes, plan):
        b = resize_bboxes(bboxes, plan['scale_factor'], plan['img_shape'],
                          self.bbox_clip_border)
        if plan['flip']:
            b = flip_bboxes(b, plan['img_shape'])
        return b

    # -- the batch ------------------------------------------------------------
    def __call__(self, images, gt_bboxes=None, gt_labels=None, rng=np.random,
                 plans=None, stream=None):
        from . import lib as L
        lib = L.get_lib()  # raises when the HIP library is missing
        if self.device.type != 'cuda':
            raise RuntimeError('DevicePipeline needs a GPU device')
        N = len(images)
        shapes = [tuple(im.shape[:2]) for im in images]
        if plans is None:
            plans = self.plan(shapes, rng)
        div = self.size_divisor or 1
        # Pad(size_divisor) per image, then collate pads to the batch maximum
        Hpad = max(_ceil_to(p['img_shape'][0], div) for p in plans)
        Wpad = max(_ceil_to(p['img_shape'][1], div) for p in plans)
        # one pinned staging buffer -> one async H2D copy for all raw images
        sizes = [h * w * 3 for h, w in shapes]
        offs = np.concatenate([[0], np.cumsum([(s + 255) // 256 * 256
                                               for s in sizes])])
        stage = torch.empty(int(offs[-1]), dtype=torch.uint8, pin_memory=True)
        for im, o, s in zip(images, offs, sizes):
            t = im if isinstance(im, torch.Tensor) else torch.from_numpy(
                np.ascontiguousarray(im))
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError('images must be uint8 HWC with 3 channels')
            stage[int(o):int(o) + s].copy_(t.reshape(-1))
        raw = stage.to(self.device, non_blocking=True)
        desc = (L.ImageT * N)()
        for i, (p, (h, w)) in enumerate(zip(plans, shapes)):
            desc[i].data = raw.data_ptr() + int(offs[i])
            desc[i].src_h, desc[i].src_w = h, w
            desc[i].new_h, desc[i].new_w = p['img_shape'][:2]
            desc[i].flip = int(p['flip'])
        dbytes = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8)
        ddesc = dbytes.pin_memory().to(self.device, non_blocking=True)
        out = torch.empty(N, 3, Hpad, Wpad, dtype=torch.float32,
                          device=self.device)
        s = stream if stream is not None else torch.cuda.current_stream(
            self.device)
        mean = (C.c_float * 3)(*self.mean.tolist())
        sinv = (C.c_float * 3)(*self.std_inv.tolist())
        L.check(lib.ld_preprocess_batch(ddesc.data_ptr(), N, Hpad, Wpad, mean,
                                        sinv, int(self.to_rgb), out.data_ptr(),
                                        s.cuda_stream), 'ld_preprocess_batch')
        metas = []
        for p in plans:
            pad_h = _ceil_to(p['img_shape'][0], div)
            pad_w = _ceil_to(p['img_shape'][1], div)
            metas.append(dict(
                ori_shape=p['ori_shape'], img_shape=p['img_shape'],
                pad_shape=(pad_h, pad_w, 3), scale_factor=p['scale_factor'],
                flip=p['flip'],
                flip_direction='horizontal' if p['flip'] else None,
                img_norm_cfg=dict(mean=self.mean, std=self.std,
                                  to_rgb=self.to_rgb)))
        result = dict(img=out, img_metas=metas)
        if gt_bboxes is not None:
            result['gt_bboxes'] = [
                torch.from_numpy(self.transform_boxes(b, p)).to(
                    self.device, non_blocking=True)
                for b, p in zip(gt_bboxes, plans)]
        if gt_labels is not None:
            result['gt_labels'] = [
                torch.as_tensor(np.asarray(l, np.int64)).to(
                    self.device, non_blocking=True) for l in gt_labels]
        return result


# ---------------------------------------------------------------------------
# samplers
# ---------------------------------------------------------------------------
class DistributedGroupSampler(torch.utils.data.Sampler):
    """mmdet/datasets/samplers/group_sampler.py:51-147.

    Images are grouped by aspect-ratio flag (``dataset.flag``: 1 when w / h > 1)
    so that a per-GPU batch pads little.  Per epoch, seeded by epoch + seed and
    identical on every rank: each group is permuted and padded (by repeating
    its own permuted order) to a multiple of samples_per_gpu * num_replicas;
    the concatenation is cut into samples_per_gpu chunks, the chunks are
    permuted, and rank r takes the r-th contiguous num_samples slice.
    """

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None,
                 rank=None, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            if num_replicas is None:
                num_replicas = dist.get_world_size() if on else 1
            if rank is None:
                rank = dist.get_rank() if on else 0
        self.dataset = dataset
        self.samples_per_gpu = samples_per_gpu
        self.num_replicas = num_replicas
        self.rank = rank
        self.epoch = 0
        self.seed = seed if seed is not None else 0
        if not hasattr(dataset, 'flag'):
            raise AttributeError('dataset needs an aspect-ratio `flag` array')
        self.flag = np.asarray(dataset.flag)
        self.group_sizes = np.bincount(self.flag)
        unit = samples_per_gpu * num_replicas
        self._padded = [int(math.ceil(int(n) / unit)) * unit
                        for n in self.group_sizes]
        self.total_size = sum(self._padded)
        self.num_samples = self.total_size // num_replicas

    def epoch_indices(self):
        """All ranks' indices for the current epoch (total_size long)."""
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        order = []
        for gid, (size, padded) in enumerate(zip(self.group_sizes,
                                                 self._padded)):
            size = int(size)
            if size == 0:
                continue
            members = np.flatnonzero(self.flag == gid)
            perm = members[torch.randperm(size, generator=g).numpy()]
            reps = np.tile(perm, padded // size + 1)[:padded]
            order.append(reps)
        order = np.concatenate(order) if order else np.zeros(0, np.int64)
        spg = self.samples_per_gpu
        chunks = torch.randperm(len(order) // spg, generator=g).numpy()
        return order.reshape(-1, spg)[chunks].reshape(-1).astype(np.int64)

    def __iter__(self):
        lo = self.num_samples * self.rank
        return iter(self.epoch_indices()[lo:lo + self.num_samples].tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class GroupSampler(torch.utils.data.Sampler):
    """mmdet/datasets/samplers/group_sampler.py:10-48 (the non-distributed
    sampler): numpy's global RNG, drawn in the reference's order."""

    def __init__(self, dataset, samples_per_gpu=1):
        if not hasattr(dataset, 'flag'):
            raise AttributeError('dataset needs an aspect-ratio `flag` array')
        self.dataset = dataset
        self.samples_per_gpu = samples_per_gpu
        self.flag = np.asarray(dataset.flag).astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(
            int(math.ceil(int(n) / samples_per_gpu)) * samples_per_gpu
            for n in self.group_sizes)

    def __iter__(self):
        spg = self.samples_per_gpu
        parts = []
        for gid, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.flatnonzero(self.flag == gid)
            np.random.shuffle(members)
            extra = int(math.ceil(int(size) / spg)) * spg - len(members)
            parts.append(np.concatenate(
                [members, np.random.choice(members, extra)]))
        flat = np.concatenate(parts)
        chunks = np.random.permutation(range(len(flat) // spg))
        return iter(flat.reshape(-1, spg)[chunks].reshape(-1).astype(
            np.int64).tolist())

    def __len__(self):
        return self.num_samples
