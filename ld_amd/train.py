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
import ctypes as C

import torch
import torch.distributed as dist

from . import layers as Y


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


_SUSPENDED = [0]


class suspend_collectives:
    """``with suspend_collectives():`` -- no gradient / scalar collective is
    issued inside (collectives_on() is False).  For steps whose result is
    thrown away on THIS rank only: the warm-up steps of a lazily captured
    hipGraph (AutoStepper).  Ranks see different padded shapes at the same
    iteration (DistributedGroupSampler), so one rank may be warming up a capture
    while its peers replay: collectives issued by the warm-up would have no
    partner."""

    def __enter__(self):
        _SUSPENDED[0] += 1

    def __exit__(self, *exc):
        _SUSPENDED[0] -= 1


def _diag_skip(kind):
    """LD_COLLECTIVES_SKIP=buckets,norm,logs -- timing diagnostics of the forced
    1-rank group only (tools/sessions): honoured when the group has ONE rank, where
    skipping a collective cannot change a result."""
    v = os.environ.get('LD_COLLECTIVES_SKIP')
    return bool(v) and kind in v.split(',') and dist.is_initialized() and \
        dist.get_world_size() == 1


def collectives_on():
    """True when gradient / scalar collectives must be issued.  With
    LD_FORCE_COLLECTIVES=1 they are issued even in a 1-rank group, which lets
    the RCCL code path be exercised on a single-GPU box."""
    if _SUSPENDED[0] or not (dist.is_available() and dist.is_initialized()):
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
                 extra_state=(), tail_bytes=8 << 20):
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
        # buckets: contiguous [start, end) ranges of the arena.  The LAST bucket
        # completes when backward does, so nothing is left to hide its
        # all-reduce behind: it is cut small (``tail_bytes``, the stem-side
        # layers) while the others stay large (``bucket_bytes``: few, big
        # messages for the point-to-point xGMI links).
        ends = [offs[i + 1] if i + 1 < len(offs) else off
                for i in range(len(order))]
        tail_from = len(order)  # first parameter of the tail bucket
        tail_cap = max(tail_bytes // 4, 0)
        if tail_cap and off > tail_cap + max(bucket_bytes // 4, 1):
            while tail_from > 1 and off - offs[tail_from - 1] <= tail_cap:
                tail_from -= 1
            if off - (offs[tail_from] if tail_from < len(order) else off) == 0:
                tail_from = len(order)
        self.buckets, start, count = [], 0, 0
        cap = max(bucket_bytes // 4, 1)
        self.bucket_of = {}
        for i, p in enumerate(order):
            end = ends[i]
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            if end - start >= cap or i + 1 == len(order) or \
                    i + 1 == tail_from:
                self.buckets.append(dict(start=start, end=end, n=count))
                start, count = end, 0
        self._ready = [0] * len(self.buckets)
        self._works = []
        self._seen = set()
        self._complete, self._next = set(), 0  # all-reduces go out in bucket order
        self._hooks = [
            p.register_post_accumulate_grad_hook(self._on_grad)
            for p in self.params
        ]
        self.enabled = True
        # optional timeline (tools/profile_step.py --buckets): a HIP event on
        # the compute stream at the moment each bucket's last gradient is
        # produced = the earliest its all-reduce can start
        self.trace = None
        # optional (bench.py at N > 1): a list that receives one (start, end) HIP
        # event pair per step around the wait for the in-flight bucket reductions
        # in finish() = the all-reduce time backward did not hide
        self.exposed = None

    def zero_grad(self):
        Y.drop_deferred()  # partials of a step that did not finish
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
        self._complete, self._next = set(), 0

    def _on_grad(self, p):
        """A parameter's gradient is complete (called by autograd's
        post-accumulate hook and/or by the direct-sink path of ld_amd.layers;
        whichever comes first counts, once per step)."""
        if not self.enabled or id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.bucket_of[id(p)]
        self._ready[b] += 1
        if self.trace is not None and self._ready[b] == self.buckets[b]['n']:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace.append((b, ev))
        if self._ready[b] == self.buckets[b]['n']:
            self._complete.add(b)
            reduce = collectives_on() and not _diag_skip('buckets')
            if not (reduce or Y.deferred_pending()):
                return
            # All-reduces go out in BUCKET ORDER on every rank, whatever order the
            # buckets complete in: a bucket that completes before an earlier one
            # (a rank on which some parameters get no gradient this step) is held
            # until the earlier ones have been issued -- by a later completion or
            # by finish().  Ranks that disagree about WHICH buckets completed
            # still issue the same sequence of collectives
            # (tests/test_ddp_real_arena.py).  In the LD step the buckets
            # complete in order, so nothing is ever held.
            todo = []
            while reduce and self._next in self._complete:
                todo.append(self._next)
                self._next += 1
            # The bucket is complete: sum the deferred partials of its weight /
            # norm gradients (layers.flush_deferred: one launch per family for
            # everything pending) and, in a multi-process job, issue its
            # all-reduce.  Weight gradients of this bucket may still run on the
            # side stream.  Round 3 made the MAIN stream wait for them here
            # (wgrad_join) -- at every bucket boundary the data-gradient chain
            # stalled behind the side stream's backlog: 56.7 -> 54.1 img/s with the
            # collectives forced in a 1-rank group (profiles/r04_bench_torchrun_
            # 1rank_forced_collectives.json), a loss every rank of an N-GPU job
            # pays.  So both are issued from the SIDE stream once that has caught
            # up with the main stream: RCCL's stream waits for the stream the
            # collective is issued from, the main stream for nobody.
            dev = self.flat_grad.device
            side = Y.wgrad_pending_stream(dev) if self.flat_grad.is_cuda else None
            if side is not None and os.environ.get('LD_BUCKET_FROM_SIDE', '1') == '1':
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    Y.flush_deferred()
                    self._all_reduce(todo)
            else:
                Y.wgrad_join()
                Y.flush_deferred()
                self._all_reduce(todo)

    def _all_reduce(self, buckets):
        for b in buckets:
            bk = self.buckets[b]
            self._works.append(dist.all_reduce(
                self.flat_grad[bk['start']:bk['end']], async_op=True))

    def finish(self):
        """Wait for the in-flight bucket reductions (sums, not yet averaged).
        Buckets whose parameters received no gradient this step, and buckets
        held back behind them, are reduced here: every rank issues the same
        collectives in the same (bucket) order."""
        Y.wgrad_join()
        Y.flush_deferred()  # parameters whose bucket never completed this step
        if collectives_on() and not _diag_skip('buckets'):
            # what was held back or never completed, still in bucket order
            self._all_reduce(range(self._next, len(self.buckets)))
            self._next = len(self.buckets)
            timed = self.exposed is not None and self.flat_grad.is_cuda
            if timed:
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
            for w in self._works:
                w.wait()
            if timed:
                b.record()
                self.exposed.append((a, b))
        self._works = []


_HWQ_CHECKED = [False]


def _check_hw_queues():
    """With a process group the step's three streams + RCCL's need more than the
    runtime's default of 4 hardware queues (ld_amd/__init__.py sets 8 at import;
    an application that initialised HIP before importing ld_amd, or exported a
    smaller value, loses the teacher / weight-gradient overlap: -4.6 % measured)."""
    if _HWQ_CHECKED[0] or not collectives_on():
        return
    _HWQ_CHECKED[0] = True
    try:
        q = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        q = 4
    if q < 8:
        import warnings
        warnings.warn(
            f'GPU_MAX_HW_QUEUES={q}: with a process group the train step\'s '
            'streams share hardware queues and stop overlapping; export '
            'GPU_MAX_HW_QUEUES=8 before the first HIP call (importing ld_amd '
            'first does it)')


def graph_queues_ok():
    """Whether a hipGraph replay of the step runs at its normal speed under the
    runtime configuration of this process: the graph executor spreads the
    captured branches over DEBUG_HIP_FORCE_GRAPH_QUEUES internal streams (4 unless
    set); with more than 4 hardware queues each gets its own queue and every
    fork / join edge of the step turns into a cross-queue dependency (bf16 replay
    15.2 -> 28 ms).  Fine with <= 4 hardware queues, or <= 2 graph streams
    (profiles/r05_graph_queues_s1.jsonl; 1 crashes the runtime, never use it)."""
    def _int(name, default):
        try:
            return int(os.environ.get(name, default))
        except ValueError:
            return default
    return _int('GPU_MAX_HW_QUEUES', 4) <= 4 or \
        _int('DEBUG_HIP_FORCE_GRAPH_QUEUES', 4) == 2


def _warn_graph_queues(what):
    if not graph_queues_ok():
        import warnings
        warnings.warn(
            f'{what}: GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES")} '
            'without DEBUG_HIP_FORCE_GRAPH_QUEUES=2 -- hipGraph replays of the '
            'train step run ~1.8x slower in this configuration (each internal '
            'graph stream lands on its own hardware queue); export '
            'DEBUG_HIP_FORCE_GRAPH_QUEUES=2 before the first HIP call (importing '
            'ld_amd first does it in a multi-process job)')


class SGDTrainer:
    """One LD training iteration = forward_train -> _parse_losses -> backward
    (with overlapped gradient all-reduce) -> SGD step."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4,
                 bucket_bytes=32 << 20, tail_bytes=8 << 20):
        _check_hw_queues()
        self.model = model
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        frozen = [p for p in model.parameters() if not p.requires_grad]
        self.arena = GradArena(list(model.parameters()), bucket_bytes,
                               extra_state=frozen + list(model.buffers()),
                               tail_bytes=tail_bytes)
        self.flat_momentum = torch.zeros_like(self.arena.flat_param)
        self.check_grads = os.environ.get('LD_CHECK_GRADS', '0') == '1'
        self.iter = 0
        self.epoch = 0
        self._hyper_dev = self._hyper_host = None

    def enable_device_hyper(self):
        """Route lr / momentum / weight decay / the 1/world gradient scale to the
        optimizer launch through a device buffer (ld_sgd_step_dev) instead of
        by-value kernel arguments -- the form a captured hipGraph needs, since a
        replay re-issues the launch with its arguments frozen.  The buffer is
        refreshed from ``self.lr`` ... before every eager step and before every
        ``GraphedStep.replay``; the arithmetic is bit-identical."""
        if self._hyper_dev is None:
            from .lossblock import PinnedRing
            dev = self.arena.flat_param.device
            self._hyper_host = PinnedRing(4, torch.float32)
            self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=dev)
            self._hyper_last = None
        self._push_hyper()

    def _push_hyper(self):
        if self._hyper_dev is None or \
                torch.cuda.is_current_stream_capturing():
            return
        vals = (float(self.lr), float(self.momentum), float(self.weight_decay),
                1.0 / _world())
        if vals == self._hyper_last:
            return  # unchanged since the last push: the device copy is current
        self._hyper_last = vals
        self._hyper_host.stage(self._hyper_dev, vals)

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
        self._push_hyper()
        Y.sgd_step(self.arena.flat_param, self.arena.flat_grad,
                   self.flat_momentum, self.lr, self.momentum,
                   self.weight_decay, 1.0 / _world(), hyper=self._hyper_dev)
        self.iter += 1
        return dict(loss=loss.detach(), log_vars=log_vars,
                    num_samples=len(data['img_metas']))


def _refuse_collectives_in_capture(what):
    """RCCL collectives inside a hipGraph capture: measured in round 5 under a
    one-rank RCCL group with the collectives forced (the only form a one-GPU box
    admits): the captured step replays bit-identical to the eager one when it
    works, but 2 of 3 runs died in ProcessGroupNCCL's WATCHDOG thread, which polls
    the events of collective work objects from another thread -- "operation not
    permitted when stream is capturing" in the default capture mode, "operation not
    permitted on an event last recorded in a capturing stream" in thread-local
    mode.  Not something this library can order, so it is refused unless
    LD_GRAPH_COLLECTIVES=1 (experiments)."""
    if collectives_on() and os.environ.get('LD_GRAPH_COLLECTIVES') != '1':
        raise RuntimeError(
            f'{what}: capturing a train step that issues RCCL collectives is '
            'refused (it races with ProcessGroupNCCL\'s watchdog thread on this '
            'stack); multi-process jobs step eagerly -- AutoStepper does that by '
            'default -- or set LD_GRAPH_COLLECTIVES=1 to try')


def _capture_mode():
    """hipStreamCaptureMode for the step captures.  Under a process group RCCL's
    watchdog THREAD polls the events of earlier collectives; in the default
    'global' mode such a query from another thread while this thread captures is an
    error that takes the process down ("operation not permitted when stream is
    capturing", seen in round 5 as a race in tests/test_gpu_graph_pg.py).
    'thread_local' restricts the checks to the capturing thread."""
    return 'thread_local' if dist.is_available() and dist.is_initialized() \
        else 'global'


class _StepList:
    """A captured step re-issued as plain stream launches (csrc/graphlist.hip,
    include/ld_hip.h "step lists"): the hipGraph is only the record.  hipGraphLaunch
    costs ~22 us of host time per node on this runtime and serialises the capture's
    branches; one C loop of hipLaunchKernel calls costs ~3-4 us per launch and keeps
    the weight gradients / the teacher on their own streams."""

    def __init__(self, graph, max_lanes=None):
        from . import lib as L
        if max_lanes is None:
            max_lanes = int(os.environ.get('LD_STEP_LIST_LANES', '4'))
        self.graph = graph  # owns the nodes' argument copies and the memory pool
        self.handle = L.get_lib().ld_step_list_build(
            C.c_void_p(graph.raw_cuda_graph()), int(max_lanes))
        if self.handle <= 0:
            raise L.LdError(f'ld_step_list_build failed ({self.handle}): the captured '
                            'step holds a node kind a launch list cannot re-issue')
        counts = (C.c_int * 8)()
        L.check(L.get_lib().ld_step_list_info(self.handle, counts), 'ld_step_list_info')
        self.info = dict(zip(('kernels', 'memcpys', 'memsets', 'ordering_nodes',
                              'lanes', 'cross_lane_waits', 'nodes'), list(counts)))

    def replay(self, device):
        from . import lib as L
        rc = L.get_lib().ld_step_list_replay(self.handle, L.stream_ptr(device))
        if rc != 0:
            f = (C.c_int * 8)()
            L.get_lib().ld_step_list_last_failure(f)
            raise L.LdError(f'ld_step_list_replay failed (hipError_t {rc}) at node '
                            f'{f[0]} of {self.info["nodes"]}: type {f[1]}, lane {f[3]}, '
                            f'details {list(f)[4:]}')

    def __del__(self):
        try:
            from . import lib as L
            if L.lib_available() and getattr(self, 'handle', 0) > 0:
                L.get_lib().ld_step_list_free(self.handle)
        except Exception:
            pass


def _new_graph(launcher):
    if launcher == 'list':
        return torch.cuda.CUDAGraph(keep_graph=True)
    if launcher != 'graph':
        raise ValueError(f"launcher must be 'graph' or 'list', got {launcher!r}")
    return torch.cuda.CUDAGraph()


class GraphedStep:
    """The steady-state train step captured ONCE into a hipGraph and replayed:
    ~750 launches per step (conv / norm / loss / optimizer kernels, the
    teacher's side stream included) become one graph launch, which takes the
    Python + ctypes enqueue cost (~13 ms per C2 step) off the critical path --
    it matters once the kernels are faster than the host (the bf16 mode).

    What a replay can change, and how (a hipGraph freezes every pointer and
    by-value argument of the launches it recorded):
      * the image batch: ``copy_inputs`` writes into the captured ``img`` buffer;
      * the ground truth, with ANY number of boxes per image up to ``max_gt``
        (the reference's batches differ every iteration,
        kd_one_stage.py:52-65): boxes / labels live in padded
        ``lossblock.StaticTargets`` buffers with the per-image count on the
        device -- the ATSS kernels loop to that run-time count;
      * the per-image ``pad_shape`` (valid anchor region): a device buffer that
        ``copy_inputs`` rewrites from the new ``img_metas``;
      * lr / momentum / weight decay: the optimizer launch reads them from a
        device buffer (``SGDTrainer.enable_device_hyper`` / ``ld_sgd_step_dev``)
        refreshed from ``trainer.lr`` ... before every replay, so lr schedules
        and ``load_state_dict`` take effect.
    What stays frozen: the padded image SHAPE (capture one GraphedStep per
    (H, W) bucket), the batch size, the precision mode, the model structure.
    Heads whose targets do not come from ``lossblock.atss_targets`` (LDFCOSHead,
    LDRetinaHead) keep the box COUNT frozen as well and ``copy_inputs`` refuses
    a different one.

    Every launch entry point of libldhip.so only enqueues (no timing, no
    synchronisation: include/ld_hip.h), which is what makes the step capturable;
    shape tuning must have happened before (ld_conv_tune_* refuse a capturing
    stream).  Under a process group the captured step would contain the bucketed
    RCCL all-reduces; that is REFUSED (see ``_refuse_collectives_in_capture``: it
    races with ProcessGroupNCCL's watchdog thread) -- multi-process jobs step
    eagerly, which round 5 made GPU-bound in bf16 as well.  ``warmup_collectives =
    False`` runs the warm-up steps without collectives (see
    ``suspend_collectives``): for captures that only this rank performs.
    """

    def __init__(self, trainer, data, warmup=2, max_gt=128,
                 warmup_collectives=True, launcher='graph'):
        """``launcher='list'``: the captured graph is re-issued by a C launch loop
        (``_StepList``) instead of hipGraphLaunch -- same launches, same buffers."""
        from . import lossblock as LB
        _refuse_collectives_in_capture('GraphedStep')
        if launcher == 'graph':
            _warn_graph_queues('GraphedStep')
        self.trainer = trainer
        self.dev = data['img'].device
        dev = data['img'].device
        n = len(data['img_metas'])
        max_gt = max(int(max_gt), max(int(b.shape[0])
                                      for b in data['gt_bboxes']))
        self.static = LB.StaticTargets(n, max_gt, dev)
        self.static.load(data['img_metas'], data['gt_bboxes'],
                         data['gt_labels'])
        head = getattr(trainer.model, 'bbox_head', None)
        self.dynamic_gt = type(head).__name__ in (
            'GFLHead', 'LDHead', 'GFocalHead', 'LDv2Head', 'ATSSGFLHead',
            'LDATSSHead')
        metas = [dict(m) for m in data['img_metas']]
        if self.dynamic_gt:
            metas[0]['ld_static_targets'] = self.static
            gtb, gtl = self.static.views()
        else:
            gtb, gtl = list(data['gt_bboxes']), list(data['gt_labels'])
        self.data = dict(img=data['img'], img_metas=metas, gt_bboxes=gtb,
                         gt_labels=gtl)
        trainer.enable_device_hyper()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # allocator / caches / weight images / the valid-region buffer
            # reach steady state on the capture stream's own key before
            # anything is recorded
            with Y.capture_warmup():
                for _ in range(max(warmup, 1)):
                    if warmup_collectives:
                        trainer.step(self.data)
                    else:  # a capture only THIS rank performs (AutoStepper)
                        with suspend_collectives():
                            trainer.step(self.data)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = _new_graph(launcher)
        with torch.cuda.graph(self.graph, stream=side,
                              capture_error_mode=_capture_mode()):
            out = trainer.step(self.data)
        self.list = _StepList(self.graph) if launcher == 'list' else None
        self._loss = out['loss']
        lv = out['log_vars']
        self._log_keys, self._log_tensor = list(lv._keys), lv._tensor
        self.num_samples = out['num_samples']

    def copy_inputs(self, data):
        """A new batch into the captured buffers: same padded image shape, any
        number of GT boxes per image up to ``max_gt``, any ``pad_shape``."""
        if data['img'].shape != self.data['img'].shape:
            raise ValueError(
                f"GraphedStep captured images of shape "
                f"{tuple(self.data['img'].shape)}, got "
                f"{tuple(data['img'].shape)}: capture one step per shape")
        if data['img'].data_ptr() != self.data['img'].data_ptr():
            self.data['img'].copy_(data['img'], non_blocking=True)
        if self.dynamic_gt:
            self.static.load(data['img_metas'], data['gt_bboxes'],
                             data['gt_labels'])
            return
        for k in ('gt_bboxes', 'gt_labels'):
            for dst, src in zip(self.data[k], data[k]):
                if dst.shape != src.shape:
                    raise ValueError(
                        'GraphedStep: this head\'s targets keep the number of '
                        'GT boxes per image frozen in the captured graph')
                dst.copy_(src, non_blocking=True)
        for a, b in zip(self.data['img_metas'], data['img_metas']):
            if tuple(a['pad_shape']) != tuple(b['pad_shape']):
                raise ValueError('GraphedStep: this head keeps pad_shape '
                                 'frozen in the captured graph')

    def replay(self):
        from .heads import LazyScalars
        self.trainer._push_hyper()
        if self.list is not None:
            self.list.replay(self.dev)
        else:
            self.graph.replay()
        self.trainer.iter += 1
        return dict(loss=self._loss,
                    log_vars=LazyScalars(self._log_keys, self._log_tensor),
                    num_samples=self.num_samples)


class PipelinedGraphedStep:
    """GraphedStep + the teacher one step ahead, as TWO hipGraphs over two
    static batch slots (round 3).  Graph k trains the student on slot k with
    the teacher outputs already sitting in slot k's static buffers and, forked
    onto the teacher stream inside the same graph, runs the frozen teacher on
    slot 1 - k into THAT slot's buffers (the teacher forward allocates from the
    graph's private pool; its outputs are copied to the static buffers: ~70 MB
    per step).  Replays alternate 0, 1, 0, ...:

        ps = PipelinedGraphedStep(trainer, batch0, batch1)
        out = ps.step(batch1)   # trains on batch0, loads batch1, teacher(batch1)
        out = ps.step(batch2)   # trains on batch1, loads batch2, teacher(batch2)

    Everything GraphedStep lets a replay change (GT count per image up to
    ``max_gt``, pad_shape, lr / momentum / weight decay) changes here too; the
    weight gradients run on their side stream inside the graph.  The results
    are bit-identical to eager steps on the same batch sequence
    (tests/test_gpu_graph.py)."""

    def __init__(self, trainer, first, second, warmup=1, max_gt=128,
                 launcher='graph'):
        from . import lossblock as LB
        _refuse_collectives_in_capture('PipelinedGraphedStep')
        if launcher == 'graph':
            _warn_graph_queues('PipelinedGraphedStep')
        self.trainer = trainer
        model = trainer.model
        if not hasattr(model, 'teacher_model') or not model.eval_teacher:
            raise ValueError('PipelinedGraphedStep needs a frozen-teacher KD '
                             'detector')
        if type(model.bbox_head).__name__ not in (
                'LDHead', 'LDv2Head', 'LDATSSHead'):
            raise NotImplementedError(
                'static (padded) targets exist for the ATSS-assigned LD heads')
        dev = first['img'].device
        self.dev = dev
        if model.teacher_stream is None:
            model.teacher_stream = Y.side_stream(dev, 'teacher')
        max_gt = max([int(max_gt)] + [int(b.shape[0]) for d in (first, second)
                                      for b in d['gt_bboxes']])
        self.slots = []
        for data in (first, second):
            n = len(data['img_metas'])
            st = LB.StaticTargets(n, max_gt, dev)
            st.load(data['img_metas'], data['gt_bboxes'], data['gt_labels'])
            metas = [dict(m) for m in data['img_metas']]
            metas[0]['ld_static_targets'] = st
            gtb, gtl = st.views()
            self.slots.append(dict(
                static=st, teacher=None,
                data=dict(img=data['img'].clone(), img_metas=metas,
                          gt_bboxes=gtb, gt_labels=gtl)))
        trainer.enable_device_hyper()
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            with Y.capture_warmup():
                for _ in range(max(warmup, 1)):
                    for sl in self.slots:
                        trainer.step(sl['data'])
            for sl in self.slots:
                tx, tout = model._teacher_forward(sl['data']['img'])
                sl['teacher'] = (tuple(t.clone() for t in tx),
                                 tuple([t.clone() for t in lvl]
                                       for lvl in tout))
        torch.cuda.current_stream(dev).wait_stream(cap)
        torch.cuda.synchronize(dev)
        self.graphs, self.outs, self.lists = [], [], []
        for k in (0, 1):
            g = _new_graph(launcher)
            with torch.cuda.graph(g, stream=cap,
                                  capture_error_mode=_capture_mode()):
                out = self._one(k)
            lv = out['log_vars']
            self.graphs.append(g)
            self.lists.append(_StepList(g) if launcher == 'list' else None)
            self.outs.append((out['loss'], list(lv._keys), lv._tensor,
                              out['num_samples']))
        self.cur = 0

    def _one(self, k):
        model = self.trainer.model
        cur, nxt = self.slots[k], self.slots[1 - k]
        main = torch.cuda.current_stream(self.dev)
        side = model.teacher_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            tx, tout = model._teacher_forward(nxt['data']['img'])
            for dst, src in zip(nxt['teacher'][0], tx):
                Y.copy_into(dst, src)
            for dl, sl in zip(nxt['teacher'][1], tout):
                for dst, src in zip(dl, sl):
                    Y.copy_into(dst, src)
        model._forced_teacher = cur['teacher']
        try:
            out = self.trainer.step(cur['data'])
        finally:
            model._forced_teacher = None
        main.wait_stream(side)
        return out

    def step(self, next_data):
        """Train on the batch loaded by the previous call (the constructor's
        ``first`` on the first call); load ``next_data`` for the following one
        and run its teacher forward inside this replay."""
        from .heads import LazyScalars
        nxt = self.slots[1 - self.cur]
        if next_data['img'].shape != nxt['data']['img'].shape:
            raise ValueError('PipelinedGraphedStep: image shape differs from '
                             'the captured one (capture one per shape)')
        nxt['data']['img'].copy_(next_data['img'], non_blocking=True)
        nxt['static'].load(next_data['img_metas'], next_data['gt_bboxes'],
                           next_data['gt_labels'])
        self.trainer._push_hyper()
        if self.lists[self.cur] is not None:
            self.lists[self.cur].replay(self.dev)
        else:
            self.graphs[self.cur].replay()
        loss, keys, tensor, ns = self.outs[self.cur]
        self.cur = 1 - self.cur
        self.trainer.iter += 1
        return dict(loss=loss, log_vars=LazyScalars(keys, tensor),
                    num_samples=ns)

class AutoStepper:
    """How a train step is enqueued (DESIGN.md sections 4 and 5):

        stepper = AutoStepper(trainer)            # mode=None: 'eager'
        for data, nxt in pairs(loader):           # nxt = the following batch
            out = stepper.step(data, next_data=nxt)

    ``step`` has the meaning of ``SGDTrainer.step``: ONE optimizer update on
    ``data``; the results are bit-identical in every mode.
      * ``'eager'`` (the default since round 5, every precision): ``trainer.step``
        with the frozen teacher of ``next_data`` one step ahead on its own stream;
      * ``'graph'`` (on request; single-process jobs): one ``GraphedStep`` per padded
        image shape (the reference's GroupSampler yields two aspect-ratio
        groups, mmdet/datasets/samplers/group_sampler.py), captured the first
        time a shape is seen.  The capture's warm-up steps are real steps on
        that batch, so the trainable state (flat parameters, momentum,
        iteration counter) is saved before and restored after: the first
        replay IS the batch's one update;
      * ``'pipelined'``: ``PipelinedGraphedStep`` (two graphs, teacher one step
        ahead inside the graph) for fixed-shape training: needs ``next_data`` on
        every call and one padded shape throughout.
    ``launcher='list'`` (round 6; 'graph' / 'pipelined' modes): the captured steps
    are re-issued by the C launch loop of ``_StepList`` instead of hipGraphLaunch --
    ~2 ms of host time per step instead of ~11, the same step time (the GPU bounds
    both precisions with one rank per GPU); what a host with few cores per GPU wants.
    """

    def __init__(self, trainer, mode=None, warmup=1, max_gt=128, max_graphs=6,
                 launcher='graph'):
        if mode is None:
            # Every precision, every job: the EAGER step with the teacher one step
            # ahead (round 5).  Round 3 made the graph path the bf16 default because
            # the eager bf16 step was host-bound then; with the frozen teacher
            # replayed from launch lists, the gradient sums inside the consumers'
            # kernels and the weight gradients on the background stream it is
            # FASTER than the replays: 162.0 img/s against 141.2 / 148.4 as one /
            # two hipGraphs (profiles/r05_bench_final_bf16_wgrad_side.json) -- a
            # graph launch costs ~19 us of host time per node on this runtime
            # (profiles/r05_graph_launch_knobs.jsonl), more than an eager launch
            # now does.  A multi-process job (BASELINE config 3's form) could not
            # capture anyway: RCCL collectives inside a capture race with
            # ProcessGroupNCCL's watchdog (_refuse_collectives_in_capture).  'graph'
            # and 'pipelined' stay available on request.
            mode = 'eager'
        if mode not in ('eager', 'graph', 'pipelined'):
            raise ValueError(f'AutoStepper: unknown mode {mode!r}')
        if launcher not in ('graph', 'list'):
            raise ValueError(f'AutoStepper: unknown launcher {launcher!r}')
        self.trainer, self.mode, self.launcher = trainer, mode, launcher
        self.warmup, self.max_gt = int(warmup), int(max_gt)
        # one graph per padded shape, at most max_graphs of them (least recently
        # used evicted: each holds a private pool with a whole step's
        # activations, and the reference's Resize(keep_ratio) + Pad(32) pipeline
        # produces dozens of padded shapes on COCO -- ADVICE r3)
        import collections
        self._graphs = collections.OrderedDict()
        self.max_graphs = max(1, int(max_graphs))
        self.evictions = 0
        if collectives_on() and mode != 'eager':
            # RCCL sets its communicator up lazily on the first collective: do
            # that here, eagerly and on every rank, never inside a capture
            t = torch.zeros(1, device=next(trainer.model.parameters()).device)
            dist.all_reduce(t)
        self._pipe = None
        self._pipe_loaded = None  # the batch object the pipeline holds for its next step
        self.captures = 0
        # a capture's warm-up runs with collectives suspended, so the
        # collective code paths (bucket all-reduces issued from the weight-
        # gradient stream, the normaliser and log reductions) would execute for
        # the first time INSIDE a capture.  The first step of a multi-process job
        # -- which every rank takes at the same iteration -- is therefore an
        # eager step with the capture's code paths, collectives on.
        self._collective_warm = not collectives_on()

    def _saved_state(self):
        tr = self.trainer
        return (tr.arena.flat_param.clone(), tr.flat_momentum.clone(), tr.iter)

    def _restore_state(self, st):
        tr = self.trainer
        tr.arena.flat_param.copy_(st[0])
        tr.flat_momentum.copy_(st[1])
        tr.iter = st[2]
        # the GEMM weight images / folded BN coefficients are refreshed AFTER
        # each optimizer launch (layers.sgd_step), for the next forward: the
        # cached ones belong to the warm-up's parameters.  Recompute them (into
        # the same buffers the captured graph reads) for the restored ones.
        Y.bump_param_generation()
        Y.refresh_params(st[0].device)

    def step(self, data, next_data=None):
        if self.mode == 'eager':
            return self.trainer.step(data, next_data=next_data)
        if not self._collective_warm:
            self._collective_warm = True
            with Y.capture_warmup():
                return self.trainer.step(data)
        if self.mode == 'graph':
            key = (tuple(data['img'].shape), len(data['img_metas']))
            g = self._graphs.get(key)
            if g is None:
                while len(self._graphs) >= self.max_graphs:
                    self._graphs.popitem(last=False)  # frees that graph's pool
                    self.evictions += 1
                st = self._saved_state()
                # the warm-up's result is discarded (state restored below) and
                # only THIS rank captures now: no collectives in it
                g = GraphedStep(self.trainer, data, warmup=self.warmup,
                                max_gt=self.max_gt, warmup_collectives=False,
                                launcher=self.launcher)
                torch.cuda.synchronize(data['img'].device)
                self._restore_state(st)
                self._graphs[key] = g
                self.captures += 1
            else:
                self._graphs.move_to_end(key)
            g.copy_inputs(data)
            return g.replay()
        # pipelined: the graph of this step runs the teacher of next_data
        if next_data is None:
            raise ValueError("AutoStepper('pipelined') needs next_data on every "
                             "call (pass the last batch again at the end)")
        if self._pipe is None:
            st = self._saved_state()
            self._pipe = PipelinedGraphedStep(self.trainer, data, next_data,
                                              warmup=self.warmup,
                                              max_gt=self.max_gt,
                                              launcher=self.launcher)
            torch.cuda.synchronize(data['img'].device)
            self._restore_state(st)
            self._pipe_loaded = data
            self.captures += 1
        if self._pipe_loaded is not data:
            raise ValueError("AutoStepper('pipelined'): this call's data must be "
                             "the previous call's next_data")
        out = self._pipe.step(next_data)
        self._pipe_loaded = next_data
        return out
