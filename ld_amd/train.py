"""Data-parallel train step (reference recipe: mmdet/apis/train.py:74-127 --
MMDistributedDataParallel + SGD + OptimizerHook -- restated MI355X-first).

One process per GPU.  Memory is laid out for the device, not inherited from
torch's per-tensor allocations:

  * every trainable parameter lives in ONE flat fp32 arena, ordered in
    *reverse* registration order (~ the order gradients become ready during
    backward); ``p.data`` and ``p.grad`` are views into the parameter / gradient
    arenas, so autograd accumulates straight into the gradient arena;
  * the gradient arena is cut into contiguous buckets; a post-accumulate hook
    counts ready tensors per bucket and launches the bucket's RCCL all-reduce
    (SUM) as soon as it is complete, on RCCL's own stream, overlapping the rest
    of backward.  xGMI is a point-to-point mesh (7 links x ~153 GB/s), so the
    default bucket is 32 MiB: few, large messages that RCCL can spread over
    all links;
  * the teacher is not a registered sub-module (kd_one_stage.py:97-108), so
    only student gradients are reduced;
  * SGD (momentum, weight decay) is ONE launch over the three arenas with the
    1/world_size averaging folded in.
The reducer itself is device-agnostic torch.distributed code (tested on gloo
with 2 CPU processes); the optimizer launch is the HIP kernel.
"""
import os

import torch
import torch.distributed as dist

from . import layers as Y


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def collectives_on():
    """True when gradient / scalar collectives must be issued.  With
    LD_FORCE_COLLECTIVES=1 they are issued even in a 1-rank group, which lets
    the RCCL code path be exercised on a single-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or \
        os.environ.get('LD_FORCE_COLLECTIVES', '0') == '1'


def broadcast_tensors(tensors, src=0):
    """Rank ``src``'s values of ``tensors`` on every rank, one collective per
    dtype (flatten -> broadcast -> scatter back).  The counterpart of the
    parameter/buffer broadcast in MMDistributedDataParallel's constructor
    (mmdet/apis/train.py:74-84): the reference seeds nothing by default, so
    without it every rank would keep its own random initialisation."""
    by_dtype = {}
    for t in tensors:
        if t.numel():
            by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            t.detach().copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class GradArena:
    """Flat parameter / gradient arenas + bucketed all-reduce.

    ``extra_state``: tensors outside the arena (frozen parameters, buffers)
    that must also start from rank 0's values."""

    def __init__(self, params, bucket_bytes=32 << 20, align=64,
                 extra_state=()):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        order = list(reversed(self.params))
        offs, off = [], 0
        for p in order:
            offs.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.order, self.offsets = order, offs
        for p, o in zip(order, offs):
            view = self.flat_param[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
            # direct sink for the backward kernels (layers._sink / _emit)
            p._ld_grad = p.grad
            p._ld_ready = self._on_grad
            p._ld_pending = 0
        if collectives_on():
            # DDP's constructor broadcast: the whole trainable arena is one
            # message; frozen parameters and buffers go packed per dtype
            dist.broadcast(self.flat_param, src=0)
            broadcast_tensors(list(extra_state), src=0)
        # buckets: contiguous [start, end) ranges of the arena
        self.buckets, start, count = [], 0, 0
        cap = max(bucket_bytes // 4, 1)
        self.bucket_of = {}
        for i, (p, o) in enumerate(zip(order, offs)):
            end = offs[i + 1] if i + 1 < len(offs) else off
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            if end - start >= cap or i + 1 == len(order):
                self.buckets.append(dict(start=start, end=end, n=count))
                start, count = end, 0
        self._ready = [0] * len(self.buckets)
        self._works = []
        self._seen = set()
        self._hooks = [
            p.register_post_accumulate_grad_hook(self._on_grad)
            for p in self.params
        ]
        self.enabled = True

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, o in zip(self.order, self.offsets):
            if p.grad is None or p.grad.data_ptr() != \
                    self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                p._ld_grad = p.grad
            p._ld_pending = 0
        self._ready = [0] * len(self.buckets)
        self._works = []
        self._seen = set()

    def _on_grad(self, p):
        """A parameter's gradient is complete (called by autograd's
        post-accumulate hook and/or by the direct-sink path of ld_amd.layers;
        whichever comes first counts, once per step)."""
        if not self.enabled or id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.bucket_of[id(p)]
        self._ready[b] += 1
        if self._ready[b] == self.buckets[b]['n'] and collectives_on():
            bk = self.buckets[b]
            self._works.append(
                dist.all_reduce(self.flat_grad[bk['start']:bk['end']],
                                async_op=True))

    def finish(self):
        """Wait for the in-flight bucket reductions (sums, not yet averaged).
        Buckets whose parameters received no gradient this step are reduced
        here so every rank issues the same collectives."""
        if collectives_on():
            for b, bk in enumerate(self.buckets):
                if self._ready[b] != bk['n']:
                    self._works.append(
                        dist.all_reduce(
                            self.flat_grad[bk['start']:bk['end']],
                            async_op=True))
            for w in self._works:
                w.wait()
        self._works = []


class SGDTrainer:
    """One LD training iteration = forward_train -> _parse_losses -> backward
    (with overlapped gradient all-reduce) -> SGD step."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4,
                 bucket_bytes=32 << 20):
        self.model = model
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        frozen = [p for p in model.parameters() if not p.requires_grad]
        self.arena = GradArena(list(model.parameters()), bucket_bytes,
                               extra_state=frozen + list(model.buffers()))
        self.flat_momentum = torch.zeros_like(self.arena.flat_param)
        self.check_grads = os.environ.get('LD_CHECK_GRADS', '0') == '1'
        self.iter = 0
        self.epoch = 0

    # -- optimizer state in torch.optim.SGD's wire format (checkpoints) -------
    def _all_params(self):
        return list(self.model.parameters())

    def state_dict(self):
        """What ``torch.optim.SGD(model.parameters(), lr, momentum,
        weight_decay).state_dict()`` would hold: one param group over ALL
        student parameters in ``model.parameters()`` order (frozen ones too, as
        mmcv's DefaultOptimizerConstructor passes them), and a
        ``momentum_buffer`` per parameter that has taken a step."""
        allp = self._all_params()
        index = {id(p): i for i, p in enumerate(allp)}
        state = {}
        if self.iter > 0 or bool(self.flat_momentum.any()):
            for p, o in zip(self.arena.order, self.arena.offsets):
                buf = self.flat_momentum[o:o + p.numel()].view_as(p)
                state[index[id(p)]] = dict(momentum_buffer=buf.detach().clone())
        group = dict(lr=self.lr, momentum=self.momentum, dampening=0,
                     weight_decay=self.weight_decay, nesterov=False,
                     params=list(range(len(allp))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1:
            raise ValueError('expected the single param group of the LD '
                             f'configs, got {len(groups)}')
        g = groups[0]
        allp = self._all_params()
        if len(g['params']) != len(allp):
            raise ValueError(f"optimizer state is for {len(g['params'])} "
                             f'parameters, the model has {len(allp)}')
        if g.get('nesterov') or g.get('dampening', 0) != 0:
            raise NotImplementedError('nesterov / dampening are not on the LD '
                                      'recipe (configs/ld/*.py: plain SGD)')
        self.lr = float(g['lr'])
        self.momentum = float(g['momentum'])
        self.weight_decay = float(g['weight_decay'])
        offset_of = {id(p): o for p, o in zip(self.arena.order,
                                              self.arena.offsets)}
        self.flat_momentum.zero_()
        for idx, st in sd['state'].items():
            p = allp[int(idx)]
            buf = st.get('momentum_buffer')
            if buf is None:
                continue
            if id(p) not in offset_of:
                raise ValueError(f'momentum for parameter {idx}, which is '
                                 'frozen in this model')
            o = offset_of[id(p)]
            self.flat_momentum[o:o + p.numel()].copy_(
                buf.reshape(-1).to(self.flat_momentum.device))

    def step(self, data, next_data=None):
        """One iteration on ``data``.  ``next_data`` (optional): the batch of
        the FOLLOWING step, already on the device -- a distillation detector
        then runs its frozen teacher on it concurrently with this step
        (KnowledgeDistillationSingleStageDetector.prefetch_teacher)."""
        if next_data is not None and hasattr(self.model, 'prefetch_teacher'):
            self.model.prefetch_teacher(next_data['img'])
        self.arena.zero_grad()
        # _parse_losses sums the loss keys with unit coefficients, so the fused
        # loss block may hand back the gradient its forward launch already
        # produced.  The promise holds for THIS backward only: the flag is
        # scoped to the step (any other consumer of the head's losses gets the
        # general rerun-with-upstream path).
        head = getattr(self.model, 'bbox_head', None)
        scoped = head is not None and hasattr(head, 'unit_upstream')
        if scoped:
            head.unit_upstream = True
        try:
            losses = self.model(**data)
            loss, log_vars = self.model._parse_losses(losses)
            loss.backward()
        finally:
            if scoped:
                head.unit_upstream = False
        if self.check_grads:
            # the single SGD launch applies weight decay / momentum to every
            # arena parameter; torch.optim.SGD skips parameters whose grad is
            # None.  Equal only if every trainable parameter got a gradient.
            missing = [i for i, p in enumerate(self.arena.order)
                       if id(p) not in self.arena._seen]
            if missing:
                raise RuntimeError(
                    f'{len(missing)} trainable parameters received no '
                    'gradient this step; the fused SGD launch would still '
                    'decay them (torch.optim.SGD would not)')
        self.arena.finish()
        Y.sgd_step(self.arena.flat_param, self.arena.flat_grad,
                   self.flat_momentum, self.lr, self.momentum,
                   self.weight_decay, 1.0 / _world())
        self.iter += 1
        return dict(loss=loss.detach(), log_vars=log_vars,
                    num_samples=len(data['img_metas']))


class GraphedStep:
    """The steady-state train step captured ONCE into a hipGraph and replayed:
    ~750 launches per step (conv / norm / loss / optimizer kernels, the
    teacher's side stream included) become one graph launch, which takes the
    Python + ctypes enqueue cost (~13 ms per C2 step, profiles/
    r01 host profile) off the critical path -- it matters once the kernels are
    faster than the host (the bf16 mode).

    Contract: ``data`` holds the STATIC input buffers; a new batch is copied
    into them (``copy_inputs``) before ``replay()``.  The shapes of the batch
    (image size, number of GT boxes per image) are frozen into the graph, as is
    the precision mode.  Every launch entry point of libldhip.so only enqueues
    (no timing, no synchronisation: include/ld_hip.h), which is what makes the
    step capturable; shape tuning must have happened before (ld_conv_tune_*
    refuse a capturing stream).  With world_size > 1 the bucketed RCCL
    all-reduces are captured like any other launch.
    """

    def __init__(self, trainer, data, warmup=2):
        self.trainer, self.data = trainer, data
        dev = data['img'].device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # allocator / caches / weight images reach steady state on the
            # capture stream's own key before anything is recorded
            for _ in range(warmup):
                trainer.step(data)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            out = trainer.step(data)
        self._loss = out['loss']
        lv = out['log_vars']
        self._log_keys, self._log_tensor = list(lv._keys), lv._tensor
        self.num_samples = out['num_samples']

    def copy_inputs(self, data):
        """New batch of the SAME shapes into the captured input buffers."""
        self.data['img'].copy_(data['img'], non_blocking=True)
        for k in ('gt_bboxes', 'gt_labels'):
            for dst, src in zip(self.data[k], data[k]):
                if dst.shape != src.shape:
                    raise ValueError(
                        'GraphedStep: the number of GT boxes per image is '
                        'frozen into the captured graph')
                dst.copy_(src, non_blocking=True)

    def replay(self):
        from .heads import LazyScalars
        self.graph.replay()
        self.trainer.iter += 1
        return dict(loss=self._loss,
                    log_vars=LazyScalars(self._log_keys, self._log_tensor),
                    num_samples=self.num_samples)
