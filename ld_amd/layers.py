"""Autograd-facing layer functions over the HIP C ABI (include/ld_hip.h).

Every arithmetic step is a libldhip.so launch on the current torch HIP stream;
torch supplies device memory, the autograd tape and nothing else.  Tensors are
handled in the (N, C, P) form: P = H*W for an ordinary NCHW map, or the sum of
the per-level H_l*W_l of a level-concatenated FPN/head tensor (``levels`` is
the tuple of (H, W) per level).
"""
import ctypes as C

import torch

from . import lib as L
from .lossblock import alloc_level_views, level_views, workspace  # noqa: F401

_PARAM_GEN = [0]


class KernelProfile:
    """Opt-in HIP-event timing of the conv launches (bench.py's roofline
    leg).  Events are recorded on the stream each kernel is launched on."""
    active = None

    def __init__(self):
        self.records = []  # (tag, flops, start_event, end_event, shape key)
        self.bytes = []
        self.rw = []       # (algorithmic bytes read, written) per launch
        self.fused = []    # (bytes read, written) by fused epilogues on top

    def __enter__(self):
        KernelProfile.active = self
        return self

    def __exit__(self, *exc):
        KernelProfile.active = None

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for tag, flops, a, b, _ in self.records:
            t, f, n = agg.get(tag, (0.0, 0.0, 0))
            agg[tag] = (t + a.elapsed_time(b) * 1e-3, f + flops, n + 1)
        return agg

    def algorithmic_bytes(self):
        """Sum over the recorded launches of the bytes a conv must move: both
        operands once + the output once, fp32 (a fused residual is not
        counted)."""
        return float(sum(self.bytes))

    def algorithmic_read_write(self):
        """The same sum split by direction: (bytes read, bytes written)."""
        return (float(sum(r for r, _ in self.rw)), float(sum(w for _, w in self.rw)))

    def fused_read_write(self):
        """What the fused epilogues move on top of the algorithmic bytes, by
        design: the residual / gradient addend they read, the raw second output
        of a conv+BN launch they write (fp32 sizes)."""
        return (float(sum(r for r, _ in self.fused)),
                float(sum(w for _, w in self.fused)))

    def by_shape(self):
        """{(tag, shape key): (seconds, flops, launches)} -- the per-layer
        table tools/profile_step.py prints."""
        torch.cuda.synchronize()
        agg = {}
        for tag, flops, a, b, key in self.records:
            t, f, n = agg.get((tag, key), (0.0, 0.0, 0))
            agg[(tag, key)] = (t + a.elapsed_time(b) * 1e-3, f + flops, n + 1)
        return agg


class _timed:

    def __init__(self, tag, d, explicit=None, fused=(0, 0)):
        # explicit = (flops, bytes read, bytes written, shape key) for a launch
        # that is not ONE conv geometry (the fused bottleneck); fused = how many
        # output-sized fp32 tensors the epilogue reads / writes IN ADDITION to
        # the conv's own operands (residual, gradient addend / raw second output)
        self.tag, self.d, self.explicit, self.fused = tag, d, explicit, fused
        self.prof = KernelProfile.active

    def __enter__(self):
        if self.prof is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.prof is not None:
            self.b.record()
            if self.explicit is not None:
                flops, rd, wr, key = self.explicit
                self.prof.records.append((self.tag, flops, self.a, self.b, key))
                self.prof.bytes.append(rd + wr)
                self.prof.rw.append((rd, wr))
                return
            d = self.d
            key = (f'{d.Cin}>{d.Cout} k{d.KH} s{d.stride} N{d.N} '
                   f'P{d.Pout} L{d.num_levels}')
            self.prof.records.append((self.tag, _conv_flops(d), self.a, self.b,
                                      key))
            xb = 4.0 * d.N * d.Cin * d.Pin
            yb = 4.0 * d.N * d.Cout * d.Pout
            wb = 4.0 * d.Cout * d.Cin * d.KH * d.KW
            self.prof.bytes.append(xb + yb + wb)
            # the same sum split by direction: what the launch reads / writes
            out = wb if 'wgrad' in self.tag else xb if 'dgrad' in self.tag else yb
            self.prof.rw.append((xb + yb + wb - out, out))
            self.prof.fused.append((self.fused[0] * out, self.fused[1] * out))


def _conv_flops(d):
    return 2.0 * d.N * d.Pout * d.Cout * d.Cin * d.KH * d.KW


def bump_param_generation():
    """Called by the optimizer after it rewrites parameters in place (outside
    torch's version counters) so cached GEMM weight images are refreshed."""
    _PARAM_GEN[0] += 1


def _dev_f32(t, name):
    L.require_device(t, torch.float32, name)
    if not t.is_contiguous():
        raise L.LdError(f'{name} must be contiguous')
    return t


def out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


_DESC_CACHE = {}


def conv_desc(N, Cin, Cout, KH, KW, stride, pad, levels):
    """(ld_conv_t, output levels) of a conv geometry; memoised (the struct is
    read-only for every caller: ~300 calls per step built it field by field)."""
    if type(levels) is not tuple or type(levels[0]) is not tuple:
        levels = tuple(tuple(int(v) for v in lv) for lv in levels)
    key = (N, Cin, Cout, KH, KW, stride, pad, levels)
    hit = _DESC_CACHE.get(key)
    if hit is None:
        if len(_DESC_CACHE) > 4096:
            _DESC_CACHE.clear()
        hit = _DESC_CACHE[key] = _conv_desc(N, Cin, Cout, KH, KW, stride, pad,
                                            tuple(levels))
    return hit


def _conv_desc(N, Cin, Cout, KH, KW, stride, pad, levels):
    d = L.ConvT()
    d.N, d.Cin, d.Cout, d.KH, d.KW = N, Cin, Cout, KH, KW
    d.stride, d.pad, d.num_levels = stride, pad, len(levels)
    pin = pout = 0
    out_levels = []
    for l, (h, w) in enumerate(levels):
        ho, wo = out_size(h, KH, stride, pad), out_size(w, KW, stride, pad)
        d.lv[l].Hin, d.lv[l].Win, d.lv[l].Hout, d.lv[l].Wout = h, w, ho, wo
        d.lv[l].off_in, d.lv[l].off_out = pin, pout
        pin += h * w
        pout += ho * wo
        out_levels.append((ho, wo))
    d.Pin, d.Pout = pin, pout
    return d, tuple(out_levels)


def levels_desc(levels):
    if type(levels) is not tuple or type(levels[0]) is not tuple:
        levels = tuple(tuple(int(v) for v in lv) for lv in levels)
    hit = _DESC_CACHE.get(levels)
    if hit is None:
        hit = _DESC_CACHE[levels] = _levels_desc(levels)
    return hit


def _levels_desc(levels):
    d = L.LevelsT()
    d.num_levels = len(levels)
    for l, (h, w) in enumerate(levels):
        d.H[l], d.W[l] = h, w
    return d


# ---------------------------------------------------------------------------
# per-step refresh of everything derived from trainable parameters
# ---------------------------------------------------------------------------
# After the optimizer step every trainable conv needs new GEMM weight images
# and every trainable BN new scale/shift: 137 + 58 launches of a few
# microseconds each on the C2 step.  Tensors that went through
# weight_images() / bn_prepare() once are remembered (weakly) and refreshed by
# ONE launch per kind from device-resident job tables.
import os  # noqa: E402
import weakref  # noqa: E402

_WT_REG, _BN_REG = {}, {}
_WT_REG_BF16 = {}
_TABLES = {}

# Matrix-operand precision of the convolutions: 'fp32' (v_mfma_f32_32x32x2_f32,
# exact fp32: BASELINE configs 1-2) or 'bf16' (v_mfma_f32_32x32x16_bf16 with
# fp32 accumulate: config 3).  Everything outside the matrix core -- master
# weights, activations, BN/GN, the loss block, gradients, SGD -- is fp32 in both
# modes (the reference's fp16 mode casts the same way: auto_fp16 around the
# nets, .float() on the head's reg output gfl_head.py:181-183, @force_fp32 on
# the loss ld_head.py:284).  Convs whose reduction is not a multiple of 16
# channels (the 3-channel stem, the data gradient of the 68-channel gfl_reg)
# stay on the fp32 kernels.
_PRECISION = [os.environ.get('LD_PRECISION', 'fp32')]


def set_precision(mode):
    if mode not in ('fp32', 'bf16'):
        raise ValueError(f'precision {mode!r}: expected fp32 or bf16')
    if mode != _PRECISION[0]:
        _PRECISION[0] = mode
        # only the images the new mode uses should be refreshed per step
        _WT_REG.clear()
        _WT_REG_BF16.clear()
        _TABLES.clear()


def get_precision():
    return _PRECISION[0]


def _use_bf16(reduction_channels):
    return _PRECISION[0] == 'bf16' and reduction_channels % 16 == 0


# bf16 mode: hand the MFMA kernels their activation operand as the bf16
# channel-blocked image (N, C/8, P, 8) instead of fp32 (N, C, P).  One HBM-bound
# conversion launch per distinct tensor (cached on the tensor: a block input
# feeds conv1 and the downsample conv, an FPN level feeds both head towers),
# then 16-byte operand loads and no conversion inside the GEMM loop
# (conv_bf16.hip conv_tile_c8_kernel).  LD_CONV_C8=0 keeps the fp32-input
# kernels.
_C8 = [os.environ.get('LD_CONV_C8', '1') == '1']


def set_c8(flag):
    _C8[0] = bool(flag)


# the C8-operand weight gradient (transpose-read kernel); LD_CONV_WGRAD_C8=0
# keeps the fp32-operand kernel
_WGRAD_C8 = [os.environ.get('LD_CONV_WGRAD_C8', '1') == '1']


def WGRAD_C8_ON():
    """Whether weight gradients take C8 operand images (a conv fed by a
    C8-only activation has no other way to get its weight gradient)."""
    return _WGRAD_C8[0] and _C8[0] and _PRECISION[0] == 'bf16' and \
        os.environ.get('LD_STUDENT_C8_ONLY', '1') == '1'


def _use_c8(reduction_channels, ksize=3, stride=1, J=1 << 30, x=None):
    """Whether a conv takes the C8 image of its activation operand.  The GEMM
    itself is faster with it everywhere (profiles/r02_kernels_c8.json: head
    tower 709 vs 495 TFLOP/s, 50x84 stages 205 vs 178), but a SEPARATE
    conversion launch only pays where the image is re-read by several taps of
    a large layer: 3x3 convs from the 100x168 stage up, and the stride-2 3x3
    convs of the backbone.
    An image that already exists (a producer wrote it, or another conv of the
    same tensor asked for it) is always used.  LD_CONV_C8=all forces it."""
    if not (_C8[0] and _PRECISION[0] == 'bf16' and
            reduction_channels % 32 == 0):
        return False
    if _C8_ALL or (x is not None and _c8_cached(x) is not None):
        return True
    return ksize >= 3 and (J >= 33600 or (stride == 2 and J >= 2048))


_C8_ALL = os.environ.get('LD_CONV_C8', '1') == 'all'
if _C8_ALL:
    _C8[0] = True


def _c8_cached(x3):
    hit = getattr(x3, '_ld_c8', None)
    if hit is None:
        # a reshape-view of the tensor the producer attached the image to (a stage
        # output handed to the neck as (N, C, H, W) and flattened again by the
        # lateral conv): same memory, same image
        b = getattr(x3, '_base', None)
        if b is not None and b.numel() == x3.numel() and \
                b.data_ptr() == x3.data_ptr() and x3.is_contiguous():
            hit = getattr(b, '_ld_c8', None)
    if hit is not None and hit[0] == (x3._version, x3.data_ptr()):
        return hit[1]
    return None


C8_STATS = dict(converted=0, reused=0)


def to_c8(x3):
    """bf16 (N, C/8, P, 8) image of an fp32 (N, C, P) tensor (cached)."""
    key = (x3._version, x3.data_ptr())
    img = _c8_cached(x3)
    if img is not None:
        C8_STATS['reused'] += 1
        return img
    if _unwritten(x3):
        raise L.LdError('to_c8: this activation exists only as a C8 image that '
                        'is no longer attached to the tensor (trunk_c8_scope)')
    C8_STATS['converted'] += 1
    N, Cc, P = x3.shape
    out = torch.empty(N * Cc * P, dtype=torch.bfloat16, device=x3.device)
    L.check(L.get_lib().ld_conv_to_c8(L.ptr(x3), N, Cc, P, L.ptr(out),
                                      L.stream_ptr(x3.device)),
            'ld_conv_to_c8')
    try:
        x3._ld_c8 = (key, out)
    except AttributeError:
        pass
    return out


class C8Act:
    """A frozen activation that exists ONLY as its bf16 C8 image (N, C/8, P, 8).

    The teacher runs under no_grad and nothing but convs read its trunk, so in
    bf16 mode its conv+BN+ReLU launches skip the fp32 output altogether and
    take the residual from the C8 image too (ld_conv_epilogue_t.y == NULL,
    residual_c8): the small stages are HBM-bound, and the fp32 master copy is
    2/3 of their bytes.  The reference's mixed-precision nets hold half
    activations end to end the same way (mmcv auto_fp16 on the backbone).
    Duck-types the few tensor members the conv plumbing touches."""
    requires_grad = False
    dtype = torch.float32  # the LOGICAL dtype of the activation

    def __init__(self, buf, shape):
        self.buf, self.shape, self.device = buf, tuple(shape), buf.device

    def reshape(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        n, c = self.shape[:2]
        rest = 1
        for v in self.shape[2:]:
            rest *= v
        if -1 in shape:
            known = 1
            for v in shape:
                known *= v if v != -1 else 1
            shape = tuple(v if v != -1 else n * c * rest // known
                          for v in shape)
        got = 1
        for v in shape[2:]:
            got *= v
        if len(shape) < 3 or shape[0] != n or shape[1] != c or got != rest:
            raise L.LdError(f'C8Act: cannot view {self.shape} as {shape} '
                            '(only the position axes may be regrouped)')
        return C8Act(self.buf, shape)

    view = reshape

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def dim(self):
        return len(self.shape)

    def float(self):
        """fp32 (N, C, ...) copy (tests / debugging only)."""
        n, c = self.shape[:2]
        return self.buf.view(n, c // 8, -1, 8).permute(0, 1, 3, 2).reshape(
            self.shape).float()


# inside this scope the fused frozen conv+BN+ReLU launches keep only the C8
# image of their output (resnet.ResNet.forward opens it for a frozen teacher)
_C8_ONLY = [False]


class c8_only_scope:

    def __enter__(self):
        self.prev, _C8_ONLY[0] = _C8_ONLY[0], True

    def __exit__(self, *exc):
        _C8_ONLY[0] = self.prev


def c8_only_available(channels):
    """Whether a frozen chain over these channel counts can run C8-only."""
    return (_C8[0] and _PRECISION[0] == 'bf16' and
            os.environ.get('LD_TEACHER_C8_ONLY', '1') == '1' and
            all(c % 32 == 0 for c in channels))


# ---- bf16 trunk between TRAINABLE blocks (round 6) ---------------------------
# Inside this scope (the KD detector opens it around the student's backbone in
# bf16 mode) a lean ConvBnActFn keeps its output z ONLY as the bf16 C8 image: the
# fp32 tensor autograd needs as the node's output is allocated but never written
# (``_ld_unwritten``), every consumer on the step takes the image -- the next conv
# as its operand, the next block's conv3 as its residual (ld_conv_epilogue_t.
# residual_c8: the identity path is then bf16 from block to block, as in the
# reference's fp16 mode, mmcv auto_fp16 on the backbone), the BN backward as its
# ReLU mask, the neck's lateral convs as their operand.  A consumer that would read
# the fp32 values raises (conv_forward_raw, to_c8, _conv_backward): no silent
# garbage.  LD_TRUNK_C8=0 keeps the fp32 copy.
_TRUNK_C8 = [False]
_TRUNK_C8_ON = [os.environ.get('LD_TRUNK_C8', '1') == '1']


class trunk_c8_scope:

    def __enter__(self):
        self.prev = _TRUNK_C8[0]
        _TRUNK_C8[0] = _TRUNK_C8_ON[0] and _PRECISION[0] == 'bf16' and _C8[0]

    def __exit__(self, *exc):
        _TRUNK_C8[0] = self.prev


def _unwritten(t):
    """Whether ``t`` (or the tensor it is a reshape view of) is an fp32
    placeholder whose values were never written (trunk_c8_scope)."""
    if not isinstance(t, torch.Tensor):
        return False
    if getattr(t, '_ld_unwritten', False):
        return True
    b = t._base
    return b is not None and getattr(b, '_ld_unwritten', False)


def _register(reg, t):
    if id(t) not in reg:
        reg[id(t)] = weakref.ref(t)
        _TABLES.clear()


def _job_table(structs, blocks_of, device):
    """ctypes job array + block->job map as device tensors."""
    first, block_job = 0, []
    for j, (st, nb) in enumerate(zip(structs, blocks_of)):
        st.first_block = first
        block_job.extend([j] * nb)
        first += nb
    arr = (type(structs[0]) * len(structs))(*structs)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return (raw.to(device), torch.tensor(block_job, dtype=torch.int32,
                                         device=device), first)


def refresh_params(device):
    """Recompute the images / coefficients of all registered trainable
    parameters for the current parameter generation."""
    lib = L.get_lib()
    gen = _PARAM_GEN[0]
    key = str(device)
    tabs = _TABLES.get(key)
    if tabs is None:
        tiled_on = os.environ.get('LD_WT_TILED', '1') == '1'

        def _wt_jobs(reg, attr, bf16):
            """jobs of one image family + blocks per job; tiled = every job is
            served by the LDS-tiled transform (ld_conv_weight_transform_batch_
            tiled: 1x1 / 3x3 convs), else the per-element batch kernel."""
            live, jobs, blocks, tiles = [], [], [], []
            for ref in list(reg.values()):
                w = ref()
                cache = getattr(w, attr, None) if w is not None else None
                if w is None or cache is None or w.device != device or \
                        cache['ident'] != (w.data_ptr(), False) or \
                        (cache['fwd'] is None and cache['bwd'] is None):
                    continue
                cout, cin, kh, kw = w.shape
                j = L.WtJobT()
                j.w = w.data_ptr()
                j.wt_fwd = cache['fwd'].data_ptr() \
                    if cache['fwd'] is not None else None
                j.wt_bwd = cache['bwd'].data_ptr() \
                    if cache['bwd'] is not None else None
                j.Cout, j.Cin, j.ntaps = cout, cin, kh * kw
                n = max(t.numel() for t in (cache['fwd'], cache['bwd'])
                        if t is not None)
                live.append((w, cache, cache['fwd'] is not None,
                             cache['bwd'] is not None))
                jobs.append(j)
                blocks.append((n + 255) // 256)
                tiles.append(lib.ld_conv_weight_transform_tiles(
                    cout, cin, kh * kw, 1 if bf16 else 0))
            tiled = tiled_on and bool(tiles) and all(t > 0 for t in tiles)
            return live, jobs, (tiles if tiled else blocks), tiled

        live_w, wjobs, wblocks, wtiled = _wt_jobs(_WT_REG, '_ld_images', False)
        live_wb, wbjobs, wbblocks, wbtiled = _wt_jobs(
            _WT_REG_BF16, '_ld_images_bf16', True)
        live_b, bjobs, bblocks = [], [], []
        for ref in list(_BN_REG.values()):
            g = ref()
            hit = getattr(g, '_ld_bn', None) if g is not None else None
            if g is None or hit is None or g.device != device:
                continue
            beta, mean, var, eps = g._ld_bn_src
            scale, shift, rstd = hit[1]
            j = L.BnJobT()
            j.gamma, j.beta, j.mean, j.var = (g.data_ptr(), beta.data_ptr(),
                                              mean.data_ptr(), var.data_ptr())
            j.scale, j.shift, j.rstd = (scale.data_ptr(), shift.data_ptr(),
                                        rstd.data_ptr())
            j.eps, j.C = eps, g.numel()
            live_b.append((g, beta, mean, var, eps))
            bjobs.append(j)
            bblocks.append((g.numel() + 255) // 256)
        tabs = dict(
            w=_job_table(wjobs, wblocks, device) if wjobs else None,
            wb=_job_table(wbjobs, wbblocks, device) if wbjobs else None,
            b=_job_table(bjobs, bblocks, device) if bjobs else None,
            live_w=live_w, live_wb=live_wb, live_b=live_b,
            tiled=dict(w=wtiled, wb=wbtiled))
        _TABLES[key] = tabs
    st = L.stream_ptr(device)
    for tkey, lkey, fn, what in (
            ('w', 'live_w', lib.ld_conv_weight_transform_batch,
             'ld_conv_weight_transform_batch'),
            ('wb', 'live_wb', lib.ld_conv_bf16_weight_transform_batch,
             'ld_conv_bf16_weight_transform_batch')):
        if tabs[tkey] is None:
            continue
        jobs, bmap, nb = tabs[tkey]
        if tabs['tiled'][tkey]:
            L.check(lib.ld_conv_weight_transform_batch_tiled(
                L.ptr(jobs), L.ptr(bmap), nb, 1 if tkey == 'wb' else 0, st),
                'ld_conv_weight_transform_batch_tiled')
        else:
            L.check(fn(L.ptr(jobs), L.ptr(bmap), nb, st), what)
        for w, cache, has_fwd, has_bwd in tabs[lkey]:
            stamp = (w._version, gen, w.data_ptr(), False)
            if cache['ident'] == stamp[2:]:
                if has_fwd and cache['fwd'] is not None:
                    cache['stamp'] = stamp
                if has_bwd and cache['bwd'] is not None:
                    cache['bwd_stamp'] = stamp
    if tabs['b'] is not None:
        jobs, bmap, nb = tabs['b']
        L.check(lib.ld_bn_prepare_batch(L.ptr(jobs), L.ptr(bmap), nb, st),
                'ld_bn_prepare_batch')
        for g, beta, mean, var, eps in tabs['live_b']:
            hit = g._ld_bn
            stamp = (g.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                     var.data_ptr(), g._version, beta._version,
                     mean._version, var._version, eps, gen)
            if hit[0][:4] == stamp[:4]:
                g._ld_bn = (stamp, hit[1])


# ---------------------------------------------------------------------------
# GEMM weight images (cached on the parameter)
# ---------------------------------------------------------------------------
def weight_images(w, need_bwd, smallc=False, bf16=False, need_fwd=True):
    """GEMM images of a conv parameter, rebuilt when the parameter changed:
    fp32 [tap][Cin][Cout] / [flipped tap][Cout][Cin] (conv.hip), or with
    ``bf16`` the [tap][K/8][C][8] bf16 images of conv_bf16.hip."""
    lib = L.get_lib()
    _dev_f32(w, 'conv weight')
    dynamic = w.requires_grad and not getattr(w, '_ld_static', False)
    stamp = (w._version, _PARAM_GEN[0] if dynamic else -1,
             w.data_ptr(), smallc)
    attr = '_ld_images_bf16' if bf16 else '_ld_images'
    reg = _WT_REG_BF16 if bf16 else _WT_REG
    cache = getattr(w, attr, None)
    if cache is None or cache['ident'] != stamp[2:]:
        cache = dict(ident=stamp[2:], stamp=None, fwd=None, bwd=None,
                     bwd_stamp=None)
        setattr(w, attr, cache)
    cout, cin, kh, kw = w.shape
    st = L.stream_ptr(w.device)
    if bf16:
        assert not smallc

        def _alloc(backward):
            n = lib.ld_conv_bf16_weight_image_elems(cout, cin, kh, kw,
                                                    backward)
            return torch.empty(n, dtype=torch.bfloat16, device=w.device)

        def _xform(fwd, bwd):
            L.check(lib.ld_conv_bf16_weight_transform(
                L.ptr(w), cout, cin, kh, kw, L.ptr(fwd), L.ptr(bwd), st),
                'ld_conv_bf16_weight_transform')
    else:
        def _alloc(backward):
            if backward:
                n = lib.ld_conv_weight_image_floats(cout, cin, kh, kw, 1)
            else:
                n = lib.ld_conv_weight_image_floats(
                    cout, cin * kh * kw if smallc else cin,
                    1 if smallc else kh, 1 if smallc else kw, 0)
            return torch.empty(n, dtype=torch.float32, device=w.device)

        def _xform(fwd, bwd):
            if smallc:
                rc = lib.ld_conv_weight_transform(
                    L.ptr(w), cout, cin * kh * kw, 1, 1, L.ptr(fwd), None, st)
            else:
                rc = lib.ld_conv_weight_transform(
                    L.ptr(w), cout, cin, kh, kw, L.ptr(fwd), L.ptr(bwd), st)
            L.check(rc, 'ld_conv_weight_transform')

    if need_fwd and cache['stamp'] != stamp:
        if cache['fwd'] is None:
            cache['fwd'] = _alloc(0)
            _TABLES.clear()
        _xform(cache['fwd'], None)
        cache['stamp'] = stamp
        if dynamic and not smallc:
            _register(reg, w)
    if need_bwd and cache['bwd_stamp'] != stamp:
        if smallc:
            raise L.LdError('small-Cin (stem) conv has no data gradient')
        if cache['bwd'] is None:
            cache['bwd'] = _alloc(1)
            _TABLES.clear()  # the refresh job of this weight gains an output
        _xform(None, cache['bwd'])
        cache['bwd_stamp'] = stamp
        if dynamic:
            _register(reg, w)
    return cache['fwd'], cache['bwd']


# ---------------------------------------------------------------------------
# explicit shape tuning (cudnn.benchmark's role, outside the launch path)
# ---------------------------------------------------------------------------
# The launch entry points only enqueue.  With tuning switched on
# (``autotune(True)`` or LD_CONV_AUTOTUNE=1) the FIRST time a conv geometry is
# seen the host calls ld_conv_tune_* once -- it times the candidate shapes on
# the buffers of that very call (idempotent launches), synchronises, and
# records the winner in the library's table, which ``save_tune_table`` writes
# out.  Off by default: the shipped table covers the benchmark shapes and
# everything else uses the library's deterministic model.
_AUTOTUNE = [os.environ.get('LD_CONV_AUTOTUNE', '0') == '1']
_TUNED = set()


def autotune(on=True):
    _AUTOTUNE[0] = bool(on)


def _tune_once(kind, d, ep_flags, call):
    if not _AUTOTUNE[0]:
        return
    key = (kind, bytes(d), ep_flags)
    if key in _TUNED:
        return
    _TUNED.add(key)
    rc = call()
    if rc not in (0, 1):
        L.check(rc, 'ld_conv_tune_' + kind)


def _epilogue(bias=None, scale=None, shift=None, residual=None, relu=False,
              y_c8=None):
    ep = L.ConvEpilogueT()
    ep.y_c8 = y_c8.data_ptr() if y_c8 is not None else None
    ep.residual_c8 = None
    if isinstance(residual, C8Act):
        L.keep(residual.buf)
        ep.residual_c8, residual = residual.buf.data_ptr(), None
    ep.y_raw = None
    ep.bias = bias.data_ptr() if bias is not None else None
    ep.scale = scale.data_ptr() if scale is not None else None
    ep.shift = shift.data_ptr() if shift is not None else None
    ep.residual = residual.data_ptr() if residual is not None else None
    ep.relu = 1 if relu else 0
    L.keep(y_c8, bias, scale, shift, residual)
    return ep


def conv_forward_raw(x3, w, stride, pad, levels, bias=None, scale=None,
                     shift=None, residual=None, relu=False, emit_c8=False,
                     c8_only=False, y_raw=None, y_raw_c8=None):
    """One implicit-GEMM launch.  Returns (y3, out_levels).  ``emit_c8``: in
    bf16 mode also write the C8 image of y from the epilogue (for a y that goes
    straight into another conv: the frozen conv+BN+ReLU chains).  ``c8_only``:
    write ONLY that image and return a C8Act.  x3 / residual may be C8Acts."""
    lib = L.get_lib()
    in8 = isinstance(x3, C8Act)
    if not in8:
        _dev_f32(x3, 'conv input')
    if _unwritten(residual):
        # a trunk activation that exists only as its C8 image (trunk_c8_scope)
        residual = C8Act(to_c8(residual), residual.shape)
    N, cin, P = x3.shape
    cout, cin_w, kh, kw = w.shape
    if cin_w != cin:
        raise L.LdError('conv: channel mismatch')
    d, out_levels = conv_desc(N, cin, cout, kh, kw, stride, pad, levels)
    if d.Pin != P:
        raise L.LdError(f'conv: input has {P} positions, levels say {d.Pin}')
    smallc = cin < 16
    bf16 = not smallc and _use_bf16(cin)
    res8 = isinstance(residual, C8Act)
    if (in8 or res8 or c8_only) and not (
            bf16 and _C8[0] and cout % 32 == 0 and (not in8 or cin % 32 == 0)):
        raise L.LdError('conv: C8-only activations need bf16 mode, the C8 path '
                        'and channel counts that are multiples of 32')
    wt_fwd, _ = weight_images(w, False, smallc, bf16)
    y3 = None if c8_only else torch.empty(
        (N, cout, d.Pout), dtype=torch.float32, device=x3.device)
    if residual is not None:
        if not res8:
            _dev_f32(residual, 'residual')
        assert tuple(residual.shape) == (N, cout, d.Pout)
    y_c8 = None
    if (emit_c8 or c8_only) and bf16 and _C8[0] and cout % 32 == 0:
        y_c8 = torch.empty(N * cout * d.Pout, dtype=torch.bfloat16,
                           device=x3.device)
    ep = _epilogue(bias, scale, shift, residual, relu, y_c8)
    if y_raw is not None:
        # second output: the conv result before the affine (ConvBnActFn)
        if smallc or c8_only or tuple(y_raw.shape) != (N, cout, d.Pout):
            raise L.LdError('conv: y_raw needs a plain fp32 output of the '
                            'same shape')
        ep.y_raw = y_raw.data_ptr()
        L.keep(y_raw)
    c8 = in8 or (bf16 and _use_c8(cin, kh, stride, N * d.Pout, x3))
    if not c8 and _unwritten(x3):
        raise L.LdError('conv: the input exists only as a C8 image '
                        '(trunk_c8_scope) and this conv takes fp32 operands')
    if y_raw_c8 is not None:
        # the pre-affine result as a bf16 C8 image (ConvBnActFn, lean form)
        if not c8 or y_raw is not None or bias is not None or cout % 8 or \
                y_raw_c8.numel() != N * cout * d.Pout or \
                y_raw_c8.dtype != torch.bfloat16:
            raise L.LdError('conv: y_raw_c8 needs the C8-operand kernel, no '
                            'bias, no fp32 y_raw and an (N*Cout*Pout) bf16 buffer')
        ep.y_raw_c8 = y_raw_c8.data_ptr()
        L.keep(y_raw_c8)
    fn = lib.ld_conv_forward_smallc if smallc else (
        lib.ld_conv_bf16_forward_c8 if c8 else
        lib.ld_conv_bf16_forward if bf16 else lib.ld_conv_forward)
    with _timed('conv_fwd_bf16' if bf16 else 'conv_fwd', d,
                fused=(int(residual is not None), int(y_raw is not None))):
        # the conversion launch is part of the conv's measured time
        xin = x3.buf if in8 else to_c8(x3) if c8 else x3
        if not smallc:
            tune = lib.ld_conv_bf16_tune_forward_c8 if c8 else \
                lib.ld_conv_bf16_tune_forward if bf16 else \
                lib.ld_conv_tune_forward
            _tune_once('c8_forward' if c8 else
                       'bf16_forward' if bf16 else 'forward', d,
                       (bias is not None, scale is not None,
                        residual is not None, bool(relu)),
                       lambda: tune(C.byref(d), L.ptr(xin), L.ptr(wt_fwd),
                                    C.byref(ep), L.ptr(y3),
                                    L.stream_ptr(x3.device)))
        L.check(fn(C.byref(d), L.ptr(xin), L.ptr(wt_fwd), C.byref(ep),
                   L.ptr(y3), L.stream_ptr(x3.device)), 'ld_conv_forward')
    if c8_only:
        return C8Act(y_c8, (N, cout, d.Pout)), out_levels
    if y_c8 is not None:
        _attach_c8(y3, y_c8)
    return y3, out_levels


# ---------------------------------------------------------------------------
# direct parameter-gradient sinks
# ---------------------------------------------------------------------------
# train.GradArena gives every trainable parameter ``_ld_grad`` (its slice of
# the flat gradient arena) and ``_ld_ready`` (the bucket bookkeeping callback).
# The backward kernels then accumulate straight into the arena and hand autograd
# ``None`` for that input, which removes one temporary + one add kernel per
# parameter per step (199 launches on the C2 step, profiles/r01_rocprof_*s22).
# Without an arena (plain ``loss.backward()``), gradients are returned as usual.
def _note_use(*params):
    for p in params:
        if p is not None and p.requires_grad and \
                getattr(p, '_ld_grad', None) is not None:
            p._ld_pending = getattr(p, '_ld_pending', 0) + 1


def _sink(p):
    if p is None or not DIRECT_GRADS[0]:
        return None
    return getattr(p, '_ld_grad', None)


def _emit(p):
    """One use of ``p`` has deposited its gradient; fire the ready callback
    after the last one."""
    p._ld_pending = getattr(p, '_ld_pending', 1) - 1
    if p._ld_pending <= 0:
        p._ld_pending = 0
        cb = getattr(p, '_ld_ready', None)
        if cb is not None:
            cb(p)


DIRECT_GRADS = [True]


# ---- weight gradients off the backward critical path (round 3) --------------
# In backward only the DATA gradient of a conv feeds the next layer; its WEIGHT
# gradient is needed by the optimizer (and the bucket's all-reduce) alone.  With
# a gradient arena the wgrad launch (+ its split-K reduce) therefore goes to a
# side stream, ordered after the kernels that produced its operands, and the
# main stream continues with the dgrad / BN-backward chain: the small-layer
# backward is a chain of 20-50 us launches at ~1 workgroup per CU, which the
# concurrent weight gradients back-fill.  Joined (main waits for the side
# stream) before a bucket's all-reduce is issued and before the optimizer step
# (train.GradArena / SGDTrainer call wgrad_join).  Same kernels, same operands:
# the result is bit-identical.  Only for gradients that go straight into the
# arena -- a gradient handed back to autograd is consumed on the main stream.
def side_stream(device, role):
    """A side stream of the train step ('teacher': the frozen teacher's forward
    one step ahead; 'wgrad': weight gradients, off the backward critical path).
    LD_SIDE_STREAM_PRIORITY = an integer gives them that HIP stream priority
    (lower number = higher priority; the main stream has 0): with a positive
    value the dispatcher prefers the critical path's kernels and lets the side
    streams fill what is left."""
    if os.environ.get('LD_SHARE_SIDE_STREAMS', '1') == '1':
        # ONE background stream for the teacher's forward and the weight
        # gradients (round 4): the critical path has a hardware queue to itself
        # and the two background jobs take turns instead of competing with it
        # and with each other.  Measured (profiles/r04_process_group_stream_
        # overlap.txt): fp32 eager 34.9 -> 34.65 ms, one hipGraph 36.7 -> 36.3 ms,
        # under a process group with 8 queues 35.05 -> 34.78 ms; with the
        # runtime held to TWO hardware queues the separate streams gave the same
        # figure (34.55) -- two lanes is what the step wants.
        key = str(device)
        if key not in _SHARED_SIDE:
            _SHARED_SIDE[key] = torch.cuda.Stream(device=device)
        return _SHARED_SIDE[key]
    pr = os.environ.get('LD_SIDE_STREAM_PRIORITY_' + role.upper(),
                        os.environ.get('LD_SIDE_STREAM_PRIORITY'))
    if pr is None:
        return torch.cuda.Stream(device=device)
    return torch.cuda.Stream(device=device, priority=int(pr))


_SHARED_SIDE = {}


_WGRAD_STREAM = [os.environ.get('LD_WGRAD_STREAM', '1') == '1']
_WGRAD_SIDE = {}
_WGRAD_PENDING = [False]
_WGRAD_FORCE = [False]
_WGRAD_BF16_EAGER = [os.environ.get('LD_WGRAD_STREAM_BF16', '1') == '1']


# ---- parameter gradients finalised per BUCKET, not per layer (round 5) ---------
# A parameter gradient is needed when its bucket is all-reduced / the optimizer
# runs -- not when its layer's backward runs.  With a gradient arena the
# cross-workgroup finalisation of the two big families is therefore deferred:
#   * weight gradients: the conv launch writes only its split partials (slabs,
#     into a buffer the weight owns), the fixed-order sum of EVERY pending weight
#     gradient runs as one ld_wgrad_reduce_batch launch when a bucket completes
#     (53 conv_wgrad_reduce launches per C2 step before, 5 now);
#   * eval-BN affine gradients: the backward launch writes its per-workgroup fp64
#     partials into a buffer the layer owns, one ld_bn_bwd_finalize_batch per bucket
#     (42 finalize launches before).
# Same additions in the same order: bit-identical gradients.  The job tables live
# on the device and are cached by content, so a steady-state step uploads nothing.
_DEFER_ON = [os.environ.get('LD_DEFER_GRADS', '1') == '1']
_DEFER_W, _DEFER_B = [], []
_DEFER_TABLES = {}
DEFER_STATS = dict(wgrad_jobs=0, bn_jobs=0, flushes=0, tables_built=0)


def _defer_buffer(owner, attr, nbytes, device):
    """A persistent scratch buffer owned by a parameter (slabs / partials live
    from the layer's backward to the bucket's flush; never shrinks).  A buffer
    born during a hipGraph capture belongs to that graph's pool: not cached."""
    t = getattr(owner, attr, None)
    if t is None or t.numel() < nbytes or t.device != device:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            try:
                setattr(owner, attr, t)
            except AttributeError:
                pass
        else:
            # only the raw pointer goes into the deferred job: hold the tensor
            # until the capture's flush has read it, or a later allocation of the
            # same capture could be handed its block
            _DEFER_CAPTURE_KEEP.append(t)
    return t


_DEFER_CAPTURE_KEEP = []  # buffers created during a hipGraph capture (process lifetime)


def _defer_table(jobs, blocks_of, device):
    key = (str(device), bytes(b''.join(bytes(j) for j in jobs)))
    hit = _DEFER_TABLES.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            # a table is an H2D copy from a host temporary: recorded into a graph
            # it would be re-read from freed memory at every replay.  The warm-up
            # steps before a capture take the capture's code paths, so this only
            # happens when they did not (loud, not a wrong gradient)
            raise L.LdError(
                'deferred-gradient job table of %d jobs was not built during the '
                'warm-up that preceded this hipGraph capture' % len(jobs))
        structs = [type(j).from_buffer_copy(bytes(j)) for j in jobs]
        hit = _job_table(structs, blocks_of, device)
        DEFER_STATS['tables_built'] += 1
        _DEFER_TABLES[key] = hit
    return hit


def flush_deferred(device=None):
    """Sum every pending deferred parameter gradient into the arena: one launch
    per family, on the CURRENT stream (the caller has ordered it after the
    launches that produced the partials)."""
    if not (_DEFER_W or _DEFER_B):
        return
    lib = L.get_lib()
    for pending, fn, what, per_block in (
            (_DEFER_W, lib.ld_wgrad_reduce_batch, 'ld_wgrad_reduce_batch', 0),
            (_DEFER_B, lib.ld_bn_bwd_finalize_batch, 'ld_bn_bwd_finalize_batch',
             16)):
        if not pending:
            continue
        jobs = [j for j, _ in pending]
        dev = pending[0][1]
        if per_block:
            blocks = [(j.C + 15) // 16 for j in jobs]
        else:
            blocks = [(j.ntaps * j.Cout * j.Cin + 1023) // 1024 for j in jobs]
        for j in jobs:
            j.first_block = 0  # set by the table builder; part of the cache key
        tab, bmap, nb = _defer_table(jobs, blocks, dev)
        prof = KernelProfile.active if not per_block else None
        if prof is not None:  # the sum belongs to the weight gradients' time
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
        L.check(fn(L.ptr(tab), L.ptr(bmap), nb, L.stream_ptr(dev)), what)
        if prof is not None:
            b.record()
            prof.records.append((
                'conv_wgrad_bf16' if _PRECISION[0] == 'bf16' else 'conv_wgrad',
                0.0, a, b, f'deferred reduce of {len(jobs)} weight gradients'))
        del pending[:]
    DEFER_STATS['flushes'] += 1


def deferred_pending():
    return bool(_DEFER_W or _DEFER_B)


def drop_deferred():
    del _DEFER_W[:]
    del _DEFER_B[:]
    drop_fan()


def drop_fan():
    """Forget fan-out deposits of a backward pass that did not finish (an
    exception skips the engine's final callbacks, so _fan_verify never ran: the
    list would then grow by every later step's activations and the check would stay
    off).  Called at the start of every step (GradArena.zero_grad)."""
    for c in _FAN_OPEN:
        c._ld_stash = None
        c._ld_fan = 0
    del _FAN_OPEN[:]


class capture_warmup:
    """Eager warm-up steps that precede a hipGraph capture must take the SAME
    code paths the capture will take (so that every cached buffer -- workspaces,
    operand images -- is created now, in ordinary memory, not during the capture
    in the graph's private pool): inside this scope the weight gradients use
    their side stream also in bf16 mode."""

    def __enter__(self):
        self.prev, _WGRAD_FORCE[0] = _WGRAD_FORCE[0], True

    def __exit__(self, *exc):
        _WGRAD_FORCE[0] = self.prev


def _wgrad_side(device):
    key = str(device)
    st = _WGRAD_SIDE.get(key)
    if st is None:
        st = side_stream(device, 'wgrad')
        _WGRAD_SIDE[key] = st
    return st


def wgrad_pending_stream(device):
    """The weight-gradient side stream of ``device`` when gradients are still
    queued on it, else None.  A bucket's all-reduce is issued FROM that stream
    (after it has waited for the main stream): the collective then follows the
    bucket's last weight gradient without making the MAIN stream -- the data
    gradient chain, i.e. the critical path of backward -- wait for the side
    stream's backlog at every bucket boundary."""
    if not _WGRAD_PENDING[0]:
        return None
    return _WGRAD_SIDE.get(str(device))


def wgrad_join(device=None):
    """Make the current stream wait for every weight gradient enqueued on the
    side stream (no-op when none is pending)."""
    if not _WGRAD_PENDING[0]:
        return
    for key, st in _WGRAD_SIDE.items():
        if device is None or key == str(device):
            torch.cuda.current_stream(st.device).wait_stream(st)
    _WGRAD_PENDING[0] = False


# ---- fan-out gradients: summed by the consumers' kernels (round 5) ------------
# Wherever an activation has several consumers (the input of a residual block:
# conv1 and the identity path, resnet.py:260-299; a stage output read by the next
# stage and an FPN lateral, fpn.py:170-176; the packed FPN levels read by both head
# towers and the loss block) the autograd engine sums the consumers' gradients
# with one elementwise ATen launch per extra consumer: 37 launches and ~0.4 ms per
# C2 step, 100-140 MB each on the big maps.  The consumers below hand their
# gradient over through this protocol instead:
#   forward   c = fan_in(ctx, i, t)      register as a participating consumer of t
#   backward  a = fan_take(c)            whatever earlier participants deposited
#             g = kernel(..., addend=a)  (conv data gradient: summed in the GEMM
#                                        epilogue, ld_conv_dgrad_acc; FPN upsample
#                                        backward: ld_upsample_add_backward_acc)
#             return fan_give(c, g)      None while participants are outstanding
#                                        (g is deposited on t), the total otherwise
# Consumers that are not participants keep going through autograd's own sum, so
# mixing is safe.  State lives on the tensor object (it dies with the graph); a
# reshape-view of a tensor resolves to its base, so x3.view(n, c, h, w) handed to
# the neck and x3 handed to the next block are one fan.  A deposit nobody
# collected by the end of the backward pass raises (a consumer's branch did not
# take part in this backward: the gradient would be silently incomplete).
_FAN_ON = [os.environ.get('LD_FAN_FUSE', '1') == '1']
_FAN_INPLACE = [os.environ.get('LD_FAN_INPLACE', '1') == '1']
_FAN_OPEN = []
FAN_STATS = dict(fused=0, fallback_adds=0)


def _fan_canon(t):
    b = t._base
    if b is not None and b.numel() == t.numel() and \
            b.data_ptr() == t.data_ptr() and b.is_contiguous() and \
            t.is_contiguous():
        return b
    return t


def fan_in(ctx, idx, t):
    """Register autograd Function ``ctx``'s input number ``idx`` (= ``t``) as a
    participant; call from ``forward`` (grad mode is off in there: whether a
    gradient is wanted is ``ctx.needs_input_grad``)."""
    if not _FAN_ON[0] or not isinstance(t, torch.Tensor) or \
            not ctx.needs_input_grad[idx] or t.grad_fn is None or \
            not t.is_contiguous():
        return None  # leaves (user inputs, parameters) stay with autograd
    c = _fan_canon(t)
    c._ld_fan = getattr(c, '_ld_fan', 0) + 1
    return c


def fan_take(c):
    if c is None:
        return None
    s = getattr(c, '_ld_stash', None)
    if s is not None:
        c._ld_stash = None
        FAN_STATS['fused'] += 1
    return s


def _fan_add(a, b):
    """a + b for two contiguous tensors of one size as the library's affine
    launch with the identity affine (a participant that could not fuse the
    deposit into its own kernel: rare paths only)."""
    FAN_STATS['fallback_adds'] += 1
    key = (str(a.device), 1)
    c = _RELU_CONSTS.get(key)
    if c is None:
        c = (torch.ones(1, device=a.device), torch.zeros(1, device=a.device))
        _RELU_CONSTS[key] = c
    out = torch.empty_like(a)
    n = a.numel()
    L.check(L.get_lib().ld_bn_act_forward(
        L.ptr(a), L.ptr(b.contiguous()), L.ptr(c[0]), L.ptr(c[1]), 1, 1, n, 0,
        L.ptr(out), L.stream_ptr(a.device)), 'ld_bn_act_forward')
    return out


def _fan_verify():
    bad = [c for c in _FAN_OPEN if getattr(c, '_ld_stash', None) is not None]
    for c in _FAN_OPEN:
        c._ld_stash = None
        c._ld_fan = 0
    del _FAN_OPEN[:]
    if bad:
        raise RuntimeError(
            f'{len(bad)} activation gradient(s) were deposited for a consumer '
            'whose backward never ran in this pass (shapes '
            f'{[tuple(c.shape) for c in bad[:4]]}): a partial backward through a '
            'fan-out; set LD_FAN_FUSE=0 for such graphs')


def fan_give(c, g):
    if c is None or g is None:
        if c is not None:
            c._ld_fan -= 1
        return g
    c._ld_fan -= 1
    s = getattr(c, '_ld_stash', None)
    if s is not None:  # a deposit this participant could not fuse
        g = _fan_add(g, s.view(g.shape))
        c._ld_stash = None
    if c._ld_fan > 0:
        c._ld_stash = g
        if not _FAN_OPEN:
            torch.autograd.Variable._execution_engine.queue_callback(
                _fan_verify)
        _FAN_OPEN.append(c)
        return None
    return g


def _conv_backward(x3, x8, w, dy, meta, params, need_x, need_w, need_b,
                   addend=None):
    """Data / weight / bias gradients of a conv (shared by ConvFn and the fused
    ConvBnActFn).  x3: the saved input tensor, or -- x8 given -- its C8Act.
    Returns (dx, dw, db); dw / db are None when they went straight into the
    gradient arena.  dy may be a C8Act (the BN backward wrote only the bf16 C8
    image of the conv's output gradient): then both gradients must be C8
    kernels, and the conv has no bias."""
    lib = L.get_lib()
    if x8 is not None:
        x3 = x8
    stride, pad, levels, has_bias = meta
    dy8 = isinstance(dy, C8Act)
    if not dy8:
        dy = dy.contiguous()
    N, cin, _ = x3.shape
    cout, _, kh, kw = w.shape
    d, _ = conv_desc(N, cin, cout, kh, kw, stride, pad, levels)
    st = L.stream_ptr(dy.device)
    dx = dw = db = None
    # bf16 + C8: the weight gradient takes both operands as C8 images
    # (ld_conv_bf16_wgrad_c8); converting dy first lets the data gradient
    # below use the same image
    c8w = need_w and _PRECISION[0] == 'bf16' and \
        _C8[0] and _WGRAD_C8[0] and cin % 32 == 0 and cout % 32 == 0
    if c8w and not dy8:
        to_c8(dy)
    if dy8 and ((need_w and not c8w) or (has_bias and need_b) or
                (need_x and not _use_bf16(cout))):
        raise L.LdError('conv backward: a C8-only output gradient needs the C8 '
                        'data- and weight-gradient kernels and no bias')
    if need_w and not c8w and _unwritten(x3):
        raise L.LdError('conv backward: the saved input exists only as a C8 '
                        'image (trunk_c8_scope); its weight gradient needs the '
                        'C8 wgrad kernel')
    if x8 is not None and not c8w:
        raise L.LdError('a C8-only activation feeds this conv: its weight '
                        'gradient needs the C8 wgrad kernel (bf16 mode, '
                        'channel counts % 32 == 0, LD_WGRAD_C8 on)')
    if need_x:
        bf16 = _use_bf16(cout)  # the data gradient reduces over Cout
        d_x, dy_x = d, dy  # descriptor / operand of the data gradient
        if not bf16 and _PAD_DGRAD[0] and _PRECISION[0] == 'bf16' and \
                not dy8 and cin >= 16 and dy.dim() == 3:
            # Cout % 16 != 0 (gfl_reg: 68 corner logits): the bf16 weight image
            # of the data gradient is zero-padded to 16 channels anyway -- pad
            # dY with zero channels to match and take the bf16 kernel instead of
            # the fp32 one (round 6: 184 -> ~80 us at the head of the backward)
            coutp = (cout + 15) // 16 * 16
            dy_x = torch.empty((N, coutp, dy.shape[2]), dtype=torch.float32,
                               device=dy.device)
            dy_x[:, :cout].copy_(dy)
            dy_x[:, cout:].zero_()
            d_x, _ = conv_desc(N, cin, coutp, kh, kw, stride, pad, levels)
            bf16 = True
        _, wt_bwd = weight_images(w, True, bf16=bf16, need_fwd=False)
        dx = torch.empty((N, cin, d.Pin), dtype=torch.float32,
                         device=dy.device)
        c8 = bf16 and (dy8 or (dy_x is dy and
                               _use_c8(cout, kh, stride, N * d.Pin, dy)))
        tune = lib.ld_conv_bf16_tune_dgrad_c8 if c8 else \
            lib.ld_conv_bf16_tune_dgrad if bf16 else \
            lib.ld_conv_tune_dgrad
        dgrad = lib.ld_conv_bf16_dgrad_c8 if c8 else \
            lib.ld_conv_bf16_dgrad if bf16 else lib.ld_conv_dgrad
        dgrad_acc = lib.ld_conv_bf16_dgrad_c8_acc if c8 else \
            lib.ld_conv_bf16_dgrad_acc if bf16 else lib.ld_conv_dgrad_acc
        with _timed('conv_dgrad_bf16' if bf16 else 'conv_dgrad', d,
                    fused=(int(addend is not None), 0)):
            dyin = dy.buf if dy8 else to_c8(dy) if c8 else dy_x
            _tune_once('c8_dgrad' if c8 else
                       'bf16_dgrad' if bf16 else 'dgrad', d_x, (),
                       lambda: tune(C.byref(d_x), L.ptr(dyin),
                                    L.ptr(wt_bwd), L.ptr(dx), st))
            if addend is not None:
                # the other consumers' gradient of this input, summed in the
                # GEMM epilogue (fan protocol above)
                if addend.numel() != dx.numel() or not addend.is_contiguous():
                    raise L.LdError('conv dgrad: addend shape mismatch')
                if stride == 2 and kh == 1 and _FAN_INPLACE[0] and \
                        getattr(addend, '_ld_fresh', False):
                    # 1x1 stride 2 reaches a quarter of the positions; the rest
                    # of dx IS the addend -- accumulate in place into the
                    # (exclusively owned) buffer another conv's data gradient
                    # deposited instead of copying it first
                    dx = addend.view(dx.shape)
                L.check(dgrad_acc(C.byref(d_x), L.ptr(dyin), L.ptr(wt_bwd),
                                  L.ptr(addend), L.ptr(dx), st),
                        'ld_conv_dgrad_acc')
            else:
                L.check(dgrad(C.byref(d_x), L.ptr(dyin), L.ptr(wt_bwd),
                              L.ptr(dx), st), 'ld_conv_dgrad')
        dx._ld_fresh = True  # nobody else holds this tensor yet
    pw, pb = params
    if need_w:
        sink = _sink(pw)
        dw = sink if sink is not None else torch.empty_like(w)
        need = lib.ld_conv_wgrad_workspace_bytes(C.byref(d))
        ws = None if sink is not None and _DEFER_ON[0] else \
            workspace(dy.device, need, 'wgrad')
        bf16 = _PRECISION[0] == 'bf16' and cin >= 16
        wgrad = lib.ld_conv_bf16_wgrad_c8 if c8w else \
            lib.ld_conv_bf16_wgrad if bf16 else lib.ld_conv_wgrad
        side = None
        # Round 3: eager bf16 steps were HOST-bound (~14 ms of enqueue per 15.5 ms
        # step) and the three extra stream calls per weight gradient cost more
        # host time than the overlap returned (15.56 -> 17.31 ms), so bf16 used
        # the side stream only inside hipGraph captures.  Round 5: the step is
        # GPU-bound (launch lists, fan protocol, deferred finalisation) and its
        # main queue is the critical path (65 % busy against 22 % on the side
        # queue, profiles/r05_queue_busy_bf16.txt): with the weight gradients
        # behind the teacher on the side stream 13.46 -> 12.42 ms per step
        # (profiles/r05_bf16_wgrad_side_stream.txt).  LD_WGRAD_STREAM_BF16=0
        # restores the old placement.  fp32: 36.88 -> 35.98 ms eager
        # (profiles/r03_wgrad_side_stream.txt).
        if sink is not None and _WGRAD_STREAM[0] and \
                dy.device.type == 'cuda' and \
                KernelProfile.active is None and (
                    _PRECISION[0] != 'bf16' or _WGRAD_FORCE[0] or
                    _WGRAD_BF16_EAGER[0] or
                    torch.cuda.is_current_stream_capturing()):
            side = _wgrad_side(dy.device)
        if not bf16 and not c8w:
            # explicit tuning writes its result with accumulate = 0: into a
            # scratch dW, never into the gradient arena
            def _tune_wgrad():
                # the tuner times every candidate plan: its own, larger workspace
                tws = torch.empty(lib.ld_conv_tune_wgrad_workspace_bytes(
                    C.byref(d)), dtype=torch.uint8, device=dy.device)
                return lib.ld_conv_tune_wgrad(
                    C.byref(d), L.ptr(x3), L.ptr(dy),
                    L.ptr(torch.empty_like(w)), L.ptr(tws), tws.numel(), st)
            _tune_once('wgrad', d, (), _tune_wgrad)
        # with a gradient arena only the split partials are computed here; their
        # sum runs once per bucket (flush_deferred)
        defer = sink is not None and _DEFER_ON[0]
        if defer and any(j.dw == sink.data_ptr() for j, _ in _DEFER_W):
            # a weight used twice: its slabs are still pending, possibly on the
            # side stream
            wgrad_join()
            flush_deferred()
        family = 2 if c8w else 1 if bf16 else 0

        def _launch(ws_, stream_ptr):
            if not defer:
                L.check(wgrad(C.byref(d), L.ptr(xw), L.ptr(dyw), L.ptr(dw),
                              0 if sink is None else 1, L.ptr(ws_),
                              ws_.numel(), stream_ptr), 'ld_conv_wgrad')
                return
            slabs = _defer_buffer(pw, '_ld_slabs', need, dy.device)
            job = L.WgradJobT()
            L.check(lib.ld_conv_wgrad_partial(
                C.byref(d), family, L.ptr(xw), L.ptr(dyw), L.ptr(slabs),
                slabs.numel(), C.byref(job), stream_ptr),
                'ld_conv_wgrad_partial')
            job.dw, job.accumulate = sink.data_ptr(), 1
            _DEFER_W.append((job, dy.device))
            DEFER_STATS['wgrad_jobs'] += 1

        with _timed('conv_wgrad_bf16' if bf16 else 'conv_wgrad', d):
            # operand images are produced on the main stream (cached ones cost
            # nothing); the wgrad itself may go to the side stream
            xw, dyw = ((x3.buf if x8 is not None else to_c8(x3)),
                       dy.buf if dy8 else to_c8(dy)) if c8w else (x3, dy)
            if side is None:
                _launch(ws, st)
            elif defer and not torch.cuda.is_current_stream_capturing():
                # no workspace, no host-side stream switch: fork the side stream
                # off the main one with ONE C call and hand the launch its raw
                # pointer (~65 weight gradients per step; the torch calls below
                # are ~25 us of host time each, this path ~6)
                sp = C.c_void_p(side.cuda_stream)
                L.check(lib.ld_stream_fork(st, sp), 'ld_stream_fork')
                _launch(ws, sp)
                for t in (xw, dyw):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(side)
                _WGRAD_PENDING[0] = True
            else:
                side.wait_stream(torch.cuda.current_stream(dy.device))
                with torch.cuda.stream(side):
                    ws2 = ws if defer else workspace(dy.device, need,
                                                     'wgrad')  # per stream
                    _launch(ws2, L.stream_ptr(dy.device))
                # the caching allocator must not hand these blocks to later
                # main-stream allocations while the side stream still reads them
                for t in (xw, dyw):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(side)
                _WGRAD_PENDING[0] = True
        if sink is not None:
            dw = None
            _emit(pw)
    if has_bias and need_b:
        sink = _sink(pb)
        if sink is not None and _DEFER_ON[0]:
            # per-channel partial sums now, finalised with the bucket's norm
            # gradients (ld_bn_bwd_finalize_batch, a job without dgamma)
            if any(j.dbeta == sink.data_ptr() for j, _ in _DEFER_B):
                # flush_deferred also reduces the pending WEIGHT slabs, which the
                # side stream may still be writing
                wgrad_join(dy.device)
                flush_deferred()
            ns = lib.ld_bias_grad_nsplit(N, cout, dy.shape[2])
            part = _defer_buffer(pb, '_ld_bias_partial', cout * ns * 16,
                                 dy.device)
            L.check(lib.ld_bias_grad_partial(L.ptr(dy), N, cout, dy.shape[2],
                                             L.ptr(part), part.numel(), st),
                    'ld_bias_grad_partial')
            job = L.BnFinJobT()
            job.partial, job.dgamma, job.dbeta = (part.data_ptr(), None,
                                                  sink.data_ptr())
            job.C, job.nsplit, job.accumulate = cout, ns, 1
            _DEFER_B.append((job, dy.device))
            DEFER_STATS['bn_jobs'] += 1
            _emit(pb)
            return dx, dw, None
        db = sink if sink is not None else \
            torch.empty(cout, dtype=torch.float32, device=dy.device)
        L.check(lib.ld_bias_grad(L.ptr(dy), N, cout, dy.shape[2],
                                 L.ptr(db), 0 if sink is None else 1, st),
                'ld_bias_grad')
        if sink is not None:
            db = None
            _emit(pb)
    return dx, dw, db


class ConvFn(torch.autograd.Function):
    """y = conv(x, w) (+ bias).  backward = MFMA dgrad + split-K wgrad."""

    @staticmethod
    def forward(ctx, x3, w, bias, stride, pad, levels):
        y3, out_levels = conv_forward_raw(x3, w, stride, pad, levels,
                                          bias=bias)
        # x3 may be a C8Act (bf16 mode: the output of the student's frozen,
        # C8-only stages): not a tensor -- keep its bf16 image, no gradient
        ctx.x8 = x3 if isinstance(x3, C8Act) else None
        if ctx.x8 is not None:
            ctx.save_for_backward(x3.buf, w)
        else:
            ctx.save_for_backward(x3, w)
        ctx.meta = (stride, pad, levels, bias is not None)
        ctx.params = (w, bias)
        ctx.fan = fan_in(ctx, 0, x3) if ctx.x8 is None else None
        _note_use(w, bias)
        return y3

    @staticmethod
    def backward(ctx, dy):
        x3, w = ctx.saved_tensors
        dx, dw, db = _conv_backward(
            x3, ctx.x8, w, dy, ctx.meta, ctx.params,
            ctx.needs_input_grad[0], ctx.needs_input_grad[1],
            ctx.needs_input_grad[2], addend=fan_take(ctx.fan))
        return fan_give(ctx.fan, dx), dw, db, None, None, None


def conv2d(x3, w, bias, stride, pad, levels):
    """Differentiable conv on an (N, C, P) tensor.  Returns (y3, out_levels)."""
    N, cin, _ = x3.shape
    cout, _, kh, kw = w.shape
    _, out_levels = conv_desc(N, cin, cout, kh, kw, stride, pad, levels)
    if torch.is_grad_enabled() and (x3.requires_grad or w.requires_grad or
                                    (bias is not None and bias.requires_grad)):
        return ConvFn.apply(x3, w, bias, stride, pad, levels), out_levels
    # no autograd: a frozen conv (teacher FPN / head) whose output feeds convs
    y3, _ = conv_forward_raw(x3, w, stride, pad, levels, bias=bias,
                             emit_c8=True)
    return y3, out_levels


# ---------------------------------------------------------------------------
# BatchNorm (eval statistics) + residual + ReLU
# ---------------------------------------------------------------------------
def bn_prepare(gamma, beta, mean, var, eps):
    """scale/shift/rstd of an eval-mode BN, cached ON the gamma tensor (a
    process-wide table keyed by address would hand a new model the
    coefficients of a freed one whose storage it happens to reuse).  Frozen BN
    (the teacher, the student's frozen stages): valid until one of the four
    tensors changes.  Trainable affine: valid for one parameter generation;
    `refresh_params` recomputes all of them in one launch after the optimizer
    step."""
    frozen = not (gamma.requires_grad or beta.requires_grad) or \
        getattr(gamma, '_ld_static', False)
    stamp = (gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
             var.data_ptr(), gamma._version, beta._version, mean._version,
             var._version, eps, -1 if frozen else _PARAM_GEN[0])
    hit = getattr(gamma, '_ld_bn', None)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    out = _bn_prepare(gamma, beta, mean, var, eps)
    gamma._ld_bn = (stamp, out)
    if not frozen:
        gamma._ld_bn_src = (beta, mean, var, eps)
        _register(_BN_REG, gamma)
    return out


def _bn_prepare(gamma, beta, mean, var, eps):
    lib = L.get_lib()
    c = gamma.numel()
    buf = torch.empty((3, c), dtype=torch.float32, device=gamma.device)
    L.check(lib.ld_bn_prepare(L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                              L.ptr(var), eps, c, L.ptr(buf[0]), L.ptr(buf[1]),
                              L.ptr(buf[2]), L.stream_ptr(gamma.device)),
            'ld_bn_prepare')
    return buf[0], buf[1], buf[2]


def _bn_act_forward_launch(x3, residual, scale, shift, N, c, P, relu, y):
    lib = L.get_lib()
    aligned = x3.data_ptr() % 16 == 0 and (
        residual is None or residual.data_ptr() % 16 == 0)
    y_c8 = _c8_side_output(y) if aligned else None
    if y_c8 is not None:
        L.check(lib.ld_bn_act_forward_c8(
            L.ptr(x3), L.ptr(residual), L.ptr(scale), L.ptr(shift), N, c, P,
            1 if relu else 0, L.ptr(y), L.ptr(y_c8),
            L.stream_ptr(x3.device)), 'ld_bn_act_forward_c8')
        _attach_c8(y, y_c8)
        return
    L.check(lib.ld_bn_act_forward(L.ptr(x3), L.ptr(residual), L.ptr(scale),
                                  L.ptr(shift), N, c, P, 1 if relu else 0,
                                  L.ptr(y), L.stream_ptr(x3.device)),
            'ld_bn_act_forward')


class BnActFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x3, gamma, beta, mean, var, eps, residual, relu):
        lib = L.get_lib()
        _dev_f32(x3, 'bn input')
        N, c, P = x3.shape
        scale, shift, rstd = bn_prepare(gamma, beta, mean, var, eps)
        y = torch.empty_like(x3)
        if residual is not None:
            _dev_f32(residual, 'bn residual')
        _bn_act_forward_launch(x3, residual, scale, shift, N, c, P, relu, y)
        ctx.save_for_backward(x3, y, scale, mean, rstd)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.params = (gamma, beta)
        ctx.fan_res = fan_in(ctx, 6, residual)
        _note_use(gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x3, y, scale, mean, rstd = ctx.saved_tensors
        ng = ctx.needs_input_grad
        dx, dres, dgamma, dbeta = _bn_act_backward(
            dy, x3, y, scale, mean, rstd, ctx.relu, ctx.params, ng[0], ng[1],
            ng[2], ctx.has_res and ng[6])
        return (dx, dgamma, dbeta, None, None, None,
                fan_give(ctx.fan_res, dres), None)


_BN_BWD_C8 = [os.environ.get('LD_BN_BWD_C8', '1') == '1']


def _bn_act_backward(dy, x3, y, scale, mean, rstd, relu, params, need_x,
                     need_g, need_b, need_res, dx_c8_only=False, c8in=None):
    """Backward of eval-BN affine (+ residual) + ReLU (shared by BnActFn and the
    fused ConvBnActFn): x3 = the BN input (conv output), y = its output.
    Returns (dx, dres, dgamma, dbeta); the parameter gradients are None when
    they went straight into the gradient arena.  ``dx_c8_only`` (bf16 mode, the
    caller's conv takes C8 operands in both of its gradients): dx is written
    ONLY as its C8 image and comes back as a C8Act -- the fp32 copy had no
    reader (round 6: 4 of the launch's 22 bytes per element).  ``c8in`` = (y image,
    x image, (N, C, P)): the lean form -- x3 / y are None, the saved activations
    are the bf16 C8 images the forward wrote (ld_bn_act_backward_c8in)."""
    lib = L.get_lib()
    dy = dy.contiguous()
    if c8in is not None:
        return _bn_act_backward_c8in(dy, c8in, scale, mean, rstd, relu, params,
                                     need_x, need_g, need_b, need_res,
                                     dx_c8_only)
    N, c, P = x3.shape
    dx8_only = bool(dx_c8_only and need_x and _BN_BWD_C8[0] and
                    N * ((P // 4 + 63) // 64) <= 256 and
                    all(t is None or t.data_ptr() % 16 == 0 for t in (dy, y, x3))
                    and _C8[0] and _PRECISION[0] == 'bf16' and c % 32 == 0 and
                    P % 4 == 0)
    dx = torch.empty_like(x3) if need_x and not dx8_only else None
    dres = torch.empty_like(x3) if need_res else None
    pg, pb = params
    sg, sb = _sink(pg), _sink(pb)
    direct = need_g and need_b and sg is not None and sb is not None
    if direct:
        dgamma, dbeta = sg, sb
    else:
        dgamma = torch.empty(c, dtype=torch.float32, device=x3.device) \
            if need_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=x3.device) \
            if need_b else None
    need = lib.ld_bn_act_backward_workspace_bytes(N, c, P)
    defer = direct and _DEFER_ON[0]
    if defer and any(j.dgamma == sg.data_ptr() for j, _ in _DEFER_B):
        # a norm used twice: its partials are still pending.  The flush also
        # reduces the pending weight slabs: wait for the side stream first
        wgrad_join(x3.device)
        flush_deferred()
    ws = _defer_buffer(pg, '_ld_bn_partial', need, x3.device) if defer else \
        workspace(x3.device, need, 'bn')
    acc = L.LD_GRAD_DEFER if defer else 1 if direct else 0
    # bf16 mode: dx goes into the conv's C8 data- / weight-gradient kernels --
    # write its C8 image from this launch instead of a to_c8 launch per conv
    dx_c8 = None
    # every tensor ld_bn_act_backward_c8 takes 16 bytes at a time must be 16-byte
    # aligned (a saved conv output or residual that is an offset view is not):
    # otherwise the plain kernel + a to_c8 launch where needed
    al16 = all(t is None or t.data_ptr() % 16 == 0 for t in (dy, y, x3, dx, dres))
    if dx8_only and not al16:  # (a fresh dres is always aligned; stay correct)
        dx, dx8_only = torch.empty_like(x3), False
        al16 = dx.data_ptr() % 16 == 0
    if dx8_only:
        dx_c8 = torch.empty(N * c * P, dtype=torch.bfloat16, device=x3.device)
    elif dx is not None and _BN_BWD_C8[0] and \
            N * ((P // 4 + 63) // 64) <= 256 and al16:
        dx_c8 = _c8_side_output(dx)
    if dx_c8 is not None:
        L.check(lib.ld_bn_act_backward_c8(
            L.ptr(dy), L.ptr(y), L.ptr(x3), L.ptr(scale), L.ptr(mean),
            L.ptr(rstd), N, c, P, 1 if relu else 0, L.ptr(dx), L.ptr(dx_c8),
            L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), acc,
            L.ptr(ws), ws.numel(), L.stream_ptr(x3.device)),
            'ld_bn_act_backward_c8')
        if dx8_only:
            dx = C8Act(dx_c8, (N, c, P))
        else:
            _attach_c8(dx, dx_c8)
    else:
        L.check(lib.ld_bn_act_backward(
            L.ptr(dy), L.ptr(y), L.ptr(x3), L.ptr(scale), L.ptr(mean),
            L.ptr(rstd), N, c, P, 1 if relu else 0, L.ptr(dx),
            L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), acc,
            L.ptr(ws), ws.numel(), L.stream_ptr(x3.device)),
            'ld_bn_act_backward')
    if defer:
        job = L.BnFinJobT()
        job.partial, job.dgamma, job.dbeta = (ws.data_ptr(), sg.data_ptr(),
                                              sb.data_ptr())
        job.C, job.accumulate = c, 1
        job.nsplit = lib.ld_bn_act_backward_nsplit(
            N, c, P, 1 if dx_c8 is not None else 0)
        _DEFER_B.append((job, x3.device))
        DEFER_STATS['bn_jobs'] += 1
    if direct:
        dgamma = dbeta = None
        _emit(pg)
        _emit(pb)
    return dx, dres, dgamma, dbeta


def _bn_act_backward_c8in(dy, c8in, scale, mean, rstd, relu, params, need_x,
                          need_g, need_b, need_res, dx_c8_only):
    """The lean BN (+ residual) + ReLU backward of bf16 mode: the saved output
    and the saved conv result are bf16 C8 images (see _bn_act_backward)."""
    lib = L.get_lib()
    y_img, x_img, (N, c, P) = c8in
    dev = dy.device
    if dy.data_ptr() % 16:  # fresh allocations are 512-byte aligned
        dy = dy.clone()
    dx = None if dx_c8_only or not need_x else torch.empty(
        (N, c, P), dtype=torch.float32, device=dev)
    dres = torch.empty((N, c, P), dtype=torch.float32, device=dev) \
        if need_res else None
    dx_c8 = torch.empty(N * c * P, dtype=torch.bfloat16, device=dev)
    pg, pb = params
    sg, sb = _sink(pg), _sink(pb)
    direct = need_g and need_b and sg is not None and sb is not None
    if direct:
        dgamma, dbeta = sg, sb
    else:
        dgamma = torch.empty(c, dtype=torch.float32, device=dev) \
            if need_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=dev) \
            if need_b else None
    need = lib.ld_bn_act_backward_workspace_bytes(N, c, P)
    defer = direct and _DEFER_ON[0]
    if defer and any(j.dgamma == sg.data_ptr() for j, _ in _DEFER_B):
        wgrad_join(dev)
        flush_deferred()
    ws = _defer_buffer(pg, '_ld_bn_partial', need, dev) if defer else \
        workspace(dev, need, 'bn')
    acc = L.LD_GRAD_DEFER if defer else 1 if direct else 0
    L.check(lib.ld_bn_act_backward_c8in(
        L.ptr(dy), L.ptr(y_img), L.ptr(x_img), L.ptr(scale), L.ptr(mean),
        L.ptr(rstd), N, c, P, 1 if relu else 0, L.ptr(dx), L.ptr(dx_c8),
        L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), acc, L.ptr(ws), ws.numel(),
        L.stream_ptr(dev)), 'ld_bn_act_backward_c8in')
    if dx is None and need_x:
        dx = C8Act(dx_c8, (N, c, P))
    elif dx is not None:
        _attach_c8(dx, dx_c8)
    if defer:
        job = L.BnFinJobT()
        job.partial, job.dgamma, job.dbeta = (ws.data_ptr(), sg.data_ptr(),
                                              sb.data_ptr())
        job.C, job.accumulate = c, 1
        job.nsplit = lib.ld_bn_act_backward_nsplit(N, c, P, 2)
        _DEFER_B.append((job, dev))
        DEFER_STATS['bn_jobs'] += 1
    if direct:
        dgamma = dbeta = None
        _emit(pg)
        _emit(pb)
    return dx, dres, dgamma, dbeta


_BN_LEAN = [os.environ.get('LD_BN_LEAN', '1') == '1']


class ConvBnActFn(torch.autograd.Function):
    """z = relu?(BN_eval(conv(x, w)) + residual) as ONE forward launch for a
    TRAINABLE conv / norm pair (round 3; VERDICT round 2, next #1b): the conv
    epilogue applies the folded eval-mode BN (resnet.py:639-648: norm_eval keeps
    the running statistics, the affine stays trainable), the residual add and
    the ReLU, and stores both z and the raw conv result (ld_conv_epilogue_t.
    y_raw) -- what round 2 did as a conv launch + a BN launch that re-read the
    conv output.  In bf16 mode the C8 image of z comes out of the same launch.
    Backward = the BN backward on (dz, z, raw) followed by the conv's data /
    weight gradients: the same kernels, in the same order, as the unfused
    pair."""

    @staticmethod
    def forward(ctx, x3, w, gamma, beta, mean, var, eps, residual, relu, stride,
                pad, levels):
        scale, shift, rstd = bn_prepare(gamma, beta, mean, var, eps)
        N = x3.shape[0]
        cout, _, kh, kw = w.shape
        cin = x3.shape[1]
        d, _ = conv_desc(N, cin, cout, kh, kw, stride, pad, levels)
        P = d.Pout
        # bf16 mode, C8-operand conv: the lean form -- the conv result before the
        # affine is kept as a bf16 C8 image (2 instead of 4 bytes per element,
        # written and read), and the backward takes the ReLU mask from the C8
        # image of z instead of reading z (round 6; LD_BN_LEAN=0: the fp32 form)
        lean = _BN_LEAN[0] and _BN_BWD_C8[0] and \
            _PRECISION[0] == 'bf16' and _C8[0] and _WGRAD_C8[0] and \
            cin % 32 == 0 and cout % 32 == 0 and P % 2 == 0 and \
            N * ((P // (4 if P % 4 == 0 else 2) + 63) // 64) <= 256 and \
            (isinstance(x3, C8Act) or
             _use_c8(cin, kh, stride, N * P, x3))
        if lean and _TRUNK_C8[0]:
            # trunk_c8_scope: z only as its C8 image; the fp32 tensor returned to
            # autograd is a placeholder nobody reads (see the scope's comment)
            raw = torch.empty(N * cout * P, dtype=torch.bfloat16,
                              device=w.device)
            z8, _ = conv_forward_raw(x3, w, stride, pad, levels, scale=scale,
                                     shift=shift, residual=residual, relu=relu,
                                     c8_only=True, y_raw_c8=raw)
            zimg = z8.buf
            z = torch.empty((N, cout, P), dtype=torch.float32, device=w.device)
            _attach_c8(z, zimg)
            z._ld_unwritten = True
        elif lean:
            raw = torch.empty(N * cout * P, dtype=torch.bfloat16,
                              device=w.device)
            z, _ = conv_forward_raw(x3, w, stride, pad, levels, scale=scale,
                                    shift=shift, residual=residual, relu=relu,
                                    emit_c8=True, y_raw_c8=raw)
            zimg = _c8_cached(z)
            if zimg is None:
                raise L.LdError('ConvBnActFn: no C8 image on the output')
        else:
            raw = torch.empty((N, cout, P), dtype=torch.float32,
                              device=w.device)
            z, _ = conv_forward_raw(x3, w, stride, pad, levels, scale=scale,
                                    shift=shift, residual=residual, relu=relu,
                                    emit_c8=True, y_raw=raw)
            zimg = z  # placeholder: same tensor, nothing extra is kept alive
        ctx.lean = lean
        ctx.oshape = (N, cout, P)
        ctx.x8 = x3 if isinstance(x3, C8Act) else None
        ctx.save_for_backward(x3.buf if ctx.x8 is not None else x3, w, raw, z,
                              scale, mean, rstd, zimg)
        ctx.meta = (stride, pad, levels, False)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.params = (w, gamma, beta)
        ctx.fan = fan_in(ctx, 0, x3) if ctx.x8 is None else None
        ctx.fan_res = fan_in(ctx, 7, residual)
        _note_use(w, gamma, beta)
        return z

    @staticmethod
    def backward(ctx, dz):
        x3, w, raw, z, scale, mean, rstd, zimg = ctx.saved_tensors
        pw, pg, pb = ctx.params
        ng = ctx.needs_input_grad
        need_conv = ng[0] or ng[1]
        # bf16 mode: both conv gradients read the C8 image of d(raw); its fp32
        # copy is then never read -- do not write it
        cout, cin = w.shape[0], w.shape[1]
        c8_dead = _DRAW_C8_ONLY[0] and _PRECISION[0] == 'bf16' and _C8[0] and \
            _WGRAD_C8[0] and cin % 32 == 0 and cout % 32 == 0 and \
            (not ng[0] or _use_bf16(cout))
        if ctx.lean:
            draw, dres, dgamma, dbeta = _bn_act_backward(
                dz, None, None, scale, mean, rstd, ctx.relu, (pg, pb),
                need_conv, ng[2], ng[3], ctx.has_res and ng[7],
                dx_c8_only=c8_dead, c8in=(zimg, raw, ctx.oshape))
        else:
            draw, dres, dgamma, dbeta = _bn_act_backward(
                dz, raw, z, scale, mean, rstd, ctx.relu, (pg, pb), need_conv,
                ng[2], ng[3], ctx.has_res and ng[7], dx_c8_only=c8_dead)
        # the identity path's gradient: deposited on the block input, where
        # conv1's data gradient (it runs later) sums it in its epilogue
        dres = fan_give(ctx.fan_res, dres)
        dx = dw = None
        if need_conv:
            dx, dw, _ = _conv_backward(x3, ctx.x8, w, draw, ctx.meta,
                                       (pw, None), ng[0], ng[1], False,
                                       addend=fan_take(ctx.fan) if ng[0]
                                       else None)
        return (fan_give(ctx.fan, dx), dw, dgamma, dbeta, None, None, None,
                dres, None, None, None, None)


_FUSE_CONV_BN = [os.environ.get('LD_FUSE_CONV_BN', '1') == '1']
_PAD_DGRAD = [os.environ.get('LD_PAD_DGRAD', '1') == '1']
_DRAW_C8_ONLY = [os.environ.get('LD_DRAW_C8_ONLY', '1') == '1']


def conv_bn_act(x3, w, gamma, beta, mean, var, eps, stride, pad, levels,
                residual=None, relu=True):
    """Differentiable conv -> eval-BN (+ residual) -> ReLU.  One fused forward
    launch (ConvBnActFn) unless LD_FUSE_CONV_BN=0 or the stem (Cin < 16, which
    is frozen in every LD config anyway); returns (z, out_levels)."""
    N, cin, _ = x3.shape
    cout, _, kh, kw = w.shape
    _, out_levels = conv_desc(N, cin, cout, kh, kw, stride, pad, levels)
    if _FUSE_CONV_BN[0] and cin >= 16:
        return ConvBnActFn.apply(x3, w, gamma, beta, mean, var, eps, residual,
                                 relu, stride, pad, levels), out_levels
    y3, _ = conv2d(x3, w, None, stride, pad, levels)
    return bn_act(y3, gamma, beta, mean, var, eps, residual, relu), out_levels


def bn_act(x3, gamma, beta, mean, var, eps, residual=None, relu=True):
    if torch.is_grad_enabled() and (x3.requires_grad or gamma.requires_grad or
                                    (residual is not None and
                                     residual.requires_grad)):
        return BnActFn.apply(x3, gamma, beta, mean, var, eps, residual, relu)
    lib = L.get_lib()
    N, c, P = x3.shape
    scale, shift, _ = bn_prepare(gamma, beta, mean, var, eps)
    y = torch.empty_like(x3)
    _bn_act_forward_launch(x3, residual, scale, shift, N, c, P, relu, y)
    return y


_RELU_CONSTS = {}


def relu(x3):
    """F.relu on an (N, C, P) tensor as the affine + ReLU launch with the
    identity affine (scale = 1 / sqrt(1 + 0) = 1 and shift = 0 exactly);
    differentiable."""
    key = (str(x3.device), int(x3.shape[1]))
    c = _RELU_CONSTS.get(key)
    if c is None:
        c = (torch.ones(key[1], device=x3.device),
             torch.zeros(key[1], device=x3.device))
        _RELU_CONSTS[key] = c
    one, zero = c
    return bn_act(x3, one, zero, zero, one, 0.0, relu=True)


def conv_bn_act_infer(x3, w, gamma, beta, mean, var, eps, stride, pad, levels,
                      residual=None, relu=True):
    """Inference-only conv -> BN(eval) -> (+residual) -> ReLU as ONE launch
    (BN folded into the conv epilogue).  Used for the frozen student stem /
    layer1 and the whole teacher."""
    scale, shift, _ = bn_prepare(gamma, beta, mean, var, eps)
    return conv_forward_raw(x3, w, stride, pad, levels, scale=scale,
                            shift=shift, residual=residual, relu=relu,
                            emit_c8=True, c8_only=_C8_ONLY[0])


# ---------------------------------------------------------------------------
# one frozen bottleneck as ONE launch (bf16 mode, C8-only trunk: the R101 teacher)
# ---------------------------------------------------------------------------
_FUSED_BLOCK = [os.environ.get('LD_FUSED_BOTTLENECK', '1') == '1']


def fused_bottleneck_available(x3, cin, mid, levels):
    """Whether ``bottleneck_c8_forward`` serves this block input."""
    if not (_FUSED_BLOCK[0] and isinstance(x3, C8Act) and _C8_ONLY[0] and
            len(levels) == 1):
        return False
    h, w = levels[0]
    return bool(L.get_lib().ld_bottleneck_c8_supported(cin, mid, h, w))


def bottleneck_c8_forward(x8, levels, convs, bns):
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + x) of an identity
    bottleneck (resnet.py:260-299) as ONE launch on C8 images
    (ld_bottleneck_c8_forward, csrc/conv_fused.hip): the mid activations stay in
    LDS.  ``convs`` / ``bns``: the block's three conv weights and (gamma, beta,
    mean, var, eps) tuples.  Bit-identical to the three fused conv+BN launches."""
    lib = L.get_lib()
    N, cin, P = x8.shape
    (h, w), = levels
    mid = convs[0].shape[0]
    b = L.BottleneckT()
    b.N, b.H, b.W, b.Cin, b.mid = N, h, w, cin, mid
    keep = []
    for i, (wt, bn) in enumerate(zip(convs, bns)):
        img, _ = weight_images(wt, False, bf16=True)
        scale, shift, _ = bn_prepare(*bn)
        if scale.data_ptr() % 16 or shift.data_ptr() % 16:
            scale, shift = scale.clone(), shift.clone()
        keep += [img, scale, shift]
        L.keep(img, scale, shift)
        setattr(b, f'w{i + 1}', img.data_ptr())
        setattr(b, f'scale{i + 1}', scale.data_ptr())
        setattr(b, f'shift{i + 1}', shift.data_ptr())
    y = torch.empty(N * cin * P, dtype=torch.bfloat16, device=x8.device)
    # bench.py's roofline leg: the three GEMMs' FLOPs; bytes = the block's input
    # and output once (fp32 sizes, as for the single convs) + the three weights
    wsz = 2 * cin * mid + 9 * mid * mid
    with _timed('conv_fwd_bf16', None,
                (2.0 * N * P * wsz, 4.0 * (N * cin * P + wsz), 4.0 * N * cin * P,
                 f'bottleneck {cin}>{mid}>{cin} N{N} P{P} L1')):
        L.check(lib.ld_bottleneck_c8_forward(C.byref(b), L.ptr(x8.buf), L.ptr(y),
                                             L.stream_ptr(x8.device)),
                'ld_bottleneck_c8_forward')
    return C8Act(y, (N, cin, P)), levels


# ---------------------------------------------------------------------------
# GroupNorm + ReLU on level-concatenated tensors
# ---------------------------------------------------------------------------
def _c8_side_output(t):
    """bf16 mode: a producer kernel may write the C8 image of its fp32 output in
    the same launch (the next conv then finds it cached on the tensor).  Returns
    the buffer to fill, or None when the geometry does not allow it."""
    N, c, P = t.shape
    if not (_C8[0] and _PRECISION[0] == 'bf16' and c % 32 == 0 and P % 4 == 0
            and t.data_ptr() % 16 == 0):
        return None
    return torch.empty(N * c * P, dtype=torch.bfloat16, device=t.device)


def _attach_c8(t, buf):
    try:
        t._ld_c8 = ((t._version, t.data_ptr()), buf)
    except AttributeError:
        pass


def _gn_forward_launch(lv, x3, gamma, beta, N, c, groups, eps, relu, y, stats):
    """y = None: write ONLY the C8 image of the output (bf16 mode; returns the
    image, or None when the geometry has no C8 form -- nothing was launched)."""
    lib = L.get_lib()
    need = lib.ld_gn_forward_workspace_bytes(C.byref(lv), N, groups)
    ws = workspace(x3.device, need, 'gn_fwd')
    if y is None:
        if not (_C8[0] and _PRECISION[0] == 'bf16' and c % 32 == 0 and
                x3.shape[2] % 4 == 0 and x3.data_ptr() % 16 == 0):
            return None
        y_c8 = torch.empty(x3.numel(), dtype=torch.bfloat16, device=x3.device)
        L.check(lib.ld_gn_forward_c8(
            C.byref(lv), L.ptr(x3), L.ptr(gamma), L.ptr(beta), N, c, groups,
            eps, 1 if relu else 0, None, L.ptr(y_c8), L.ptr(stats[0]),
            L.ptr(stats[1]), L.ptr(ws), ws.numel(),
            L.stream_ptr(x3.device)), 'ld_gn_forward_c8')
        return y_c8
    y_c8 = _c8_side_output(y) if x3.data_ptr() % 16 == 0 else None
    if y_c8 is not None:
        L.check(lib.ld_gn_forward_c8(
            C.byref(lv), L.ptr(x3), L.ptr(gamma), L.ptr(beta), N, c, groups,
            eps, 1 if relu else 0, L.ptr(y), L.ptr(y_c8), L.ptr(stats[0]),
            L.ptr(stats[1]), L.ptr(ws), ws.numel(),
            L.stream_ptr(x3.device)), 'ld_gn_forward_c8')
        _attach_c8(y, y_c8)
        return
    L.check(lib.ld_gn_forward(C.byref(lv), L.ptr(x3), L.ptr(gamma),
                              L.ptr(beta), N, c, groups, eps,
                              1 if relu else 0, L.ptr(y), L.ptr(stats[0]),
                              L.ptr(stats[1]), L.ptr(ws), ws.numel(),
                              L.stream_ptr(x3.device)), 'ld_gn_forward')


_GN_LEAN = [os.environ.get('LD_GN_LEAN', '1') == '1']
_GN_YSKIP = [os.environ.get('LD_GN_YSKIP', '1') == '1']


def _gn_backward(dy, x3, y, gamma, beta, stats, groups, levels, relu, params,
                 need_g, need_b, dx_c8_only=False, need_lean=False):
    """Backward of GroupNorm (+ ReLU) on a level-concatenated tensor (shared by
    GnActFn and the fused ConvGnActFn).  Returns (dx, dgamma, dbeta); the
    parameter gradients are None when they went into the gradient arena.
    bf16 mode (C8 side output possible): the lean kernels -- the ReLU mask is
    recomputed from x instead of reading y back (bit-identical, LD_GN_LEAN=0 for
    the old form) and, ``dx_c8_only``, dx exists only as its C8 image and comes
    back as a C8Act."""
    lib = L.get_lib()
    dy = dy.contiguous()
    N, c, P = x3.shape
    lv = levels_desc(levels)
    pg, pb = params
    sg, sb = _sink(pg), _sink(pb)
    direct = need_g and need_b and sg is not None and sb is not None
    if direct:
        dgamma, dbeta = sg, sb
    else:
        dgamma = torch.empty(c, dtype=torch.float32, device=x3.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x3.device)
    need = lib.ld_gn_backward_workspace_bytes(C.byref(lv), N, c)
    ws = workspace(x3.device, need, 'gn')
    aligned = all(t.data_ptr() % 16 == 0 for t in (dy, y, x3))
    c8_ok = aligned and _C8[0] and _PRECISION[0] == 'bf16' and c % 32 == 0 and \
        P % 4 == 0
    lean = c8_ok and _GN_LEAN[0] and os.environ.get('LD_NN_OLD') != '1'
    if need_lean and not lean:
        raise L.LdError('GroupNorm backward: the forward kept y only as its C8 '
                        'image, the kernels that do not read y are not '
                        'available for these operands')
    dx8_only = bool(lean and dx_c8_only)
    dx = None if dx8_only else torch.empty_like(x3)
    dx_c8 = torch.empty(N * c * P, dtype=torch.bfloat16, device=x3.device) \
        if dx8_only else _c8_side_output(dx) if c8_ok else None
    if lean:
        L.check(lib.ld_gn_backward_c8_lean(
            C.byref(lv), L.ptr(dy), L.ptr(x3), L.ptr(gamma), L.ptr(beta),
            L.ptr(stats[0]), L.ptr(stats[1]), N, c, groups,
            1 if relu else 0, L.ptr(dx), L.ptr(dx_c8), L.ptr(dgamma),
            L.ptr(dbeta), 1 if direct else 0, L.ptr(ws), ws.numel(),
            L.stream_ptr(x3.device)), 'ld_gn_backward_c8_lean')
        if dx8_only:
            dx = C8Act(dx_c8, (N, c, P))
        else:
            _attach_c8(dx, dx_c8)
    elif dx_c8 is not None:
        L.check(lib.ld_gn_backward_c8(
            C.byref(lv), L.ptr(dy), L.ptr(y), L.ptr(x3), L.ptr(gamma),
            L.ptr(stats[0]), L.ptr(stats[1]), N, c, groups,
            1 if relu else 0, L.ptr(dx), L.ptr(dx_c8), L.ptr(dgamma),
            L.ptr(dbeta), 1 if direct else 0, L.ptr(ws), ws.numel(),
            L.stream_ptr(x3.device)), 'ld_gn_backward_c8')
        _attach_c8(dx, dx_c8)
    else:
        L.check(lib.ld_gn_backward(
            C.byref(lv), L.ptr(dy), L.ptr(y), L.ptr(x3), L.ptr(gamma),
            L.ptr(stats[0]), L.ptr(stats[1]), N, c, groups,
            1 if relu else 0, L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta),
            1 if direct else 0, L.ptr(ws), ws.numel(),
            L.stream_ptr(x3.device)), 'ld_gn_backward')
    if direct:
        dgamma = dbeta = None
        _emit(pg)
        _emit(pb)
    return dx, dgamma, dbeta


class GnActFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x3, gamma, beta, groups, eps, levels, relu):
        _dev_f32(x3, 'gn input')
        N, c, P = x3.shape
        lv = levels_desc(levels)
        y = torch.empty_like(x3)
        stats = torch.empty((2, N, groups, len(levels)), dtype=torch.float32,
                            device=x3.device)
        _gn_forward_launch(lv, x3, gamma, beta, N, c, groups, eps, relu, y,
                           stats)
        ctx.save_for_backward(x3, y, gamma, beta, stats)
        ctx.meta = (groups, levels, relu)
        ctx.params = (gamma, beta)
        _note_use(gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x3, y, gamma, beta, stats = ctx.saved_tensors
        groups, levels, relu = ctx.meta
        dx, dgamma, dbeta = _gn_backward(
            dy, x3, y, gamma, beta, stats, groups, levels, relu, ctx.params,
            ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dgamma, dbeta, None, None, None, None


class ConvGnActFn(torch.autograd.Function):
    """y = relu?(GroupNorm(conv(x, w))) for a TRAINABLE bias-free conv / GN pair
    (the GFL head towers, gfl_head.py:102-133) as ONE autograd node (round 6, bf16
    mode).  The launches are those of ConvFn + GnActFn, in the same order, on the
    same operands; what the fusion buys is in the backward: the gradient of the
    conv output never leaves this node, so it exists only as the bf16 C8 image
    both conv gradients read (no fp32 copy: 4 of the GN backward's bytes per
    element), and the conv output is not an autograd tensor."""

    @staticmethod
    def forward(ctx, x3, w, gamma, beta, groups, eps, stride, pad, levels, relu,
                c8_out=False):
        raw, out_levels = conv_forward_raw(x3, w, stride, pad, levels)
        N, c, P = raw.shape
        lv = levels_desc(out_levels)
        stats = torch.empty((2, N, groups, len(out_levels)),
                            dtype=torch.float32, device=raw.device)
        # ``c8_out`` (the caller's next layer is another conv of the stack): the
        # lean backward does not read y and the next conv takes the C8 image --
        # the fp32 output is then a placeholder that is never written
        # (``_ld_unwritten``, see trunk_c8_scope)
        img = None
        if c8_out and _TRUNK_C8_ON[0] and _GN_YSKIP[0] and _GN_LEAN[0] and \
                os.environ.get('LD_NN_OLD') != '1':
            img = _gn_forward_launch(lv, raw, gamma, beta, N, c, groups, eps,
                                     relu, None, stats)
        y = torch.empty_like(raw)
        if img is not None:
            _attach_c8(y, img)
            y._ld_unwritten = True
        else:
            _gn_forward_launch(lv, raw, gamma, beta, N, c, groups, eps, relu, y,
                               stats)
        ctx.y_unwritten = img is not None
        ctx.x8 = x3 if isinstance(x3, C8Act) else None
        ctx.save_for_backward(x3.buf if ctx.x8 is not None else x3, w, raw, y,
                              gamma, beta, stats)
        ctx.meta = (stride, pad, levels, False)
        ctx.gn = (groups, out_levels, relu)
        ctx.params = (w, gamma, beta)
        ctx.fan = fan_in(ctx, 0, x3) if ctx.x8 is None else None
        _note_use(w, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x3, w, raw, y, gamma, beta, stats = ctx.saved_tensors
        pw, pg, pb = ctx.params
        groups, out_levels, relu = ctx.gn
        ng = ctx.needs_input_grad
        cout, cin = w.shape[0], w.shape[1]
        c8_dead = _DRAW_C8_ONLY[0] and _PRECISION[0] == 'bf16' and _C8[0] and \
            _WGRAD_C8[0] and cin % 32 == 0 and cout % 32 == 0 and \
            (not ng[0] or _use_bf16(cout))
        draw, dgamma, dbeta = _gn_backward(
            dy, raw, y, gamma, beta, stats, groups, out_levels, relu, (pg, pb),
            ng[2], ng[3], dx_c8_only=c8_dead, need_lean=ctx.y_unwritten)
        dx = dw = None
        if ng[0] or ng[1]:
            dx, dw, _ = _conv_backward(x3, ctx.x8, w, draw, ctx.meta,
                                       (pw, None), ng[0], ng[1], False,
                                       addend=fan_take(ctx.fan) if ng[0]
                                       else None)
        return (fan_give(ctx.fan, dx), dw, dgamma, dbeta, None, None, None,
                None, None, None, None)


_FUSE_CONV_GN = [os.environ.get('LD_FUSE_CONV_GN', '1') == '1']


def conv_gn_act(x3, w, gamma, beta, groups, eps, stride, pad, levels, relu=True,
                c8_only=False):
    """Bias-free conv -> GroupNorm -> ReLU; returns (y3, out_levels).  bf16 mode:
    one autograd node when something trains (ConvGnActFn); without a gradient
    ``c8_only`` asks for the output as a C8Act (a caller whose next layer is
    another C8 conv).  The conv2d + gn_act pair otherwise (fp32 mode,
    LD_FUSE_CONV_GN=0)."""
    N, cin, _ = x3.shape
    cout, _, kh, kw = w.shape
    _, out_levels = conv_desc(N, cin, cout, kh, kw, stride, pad, levels)
    grad = torch.is_grad_enabled() and (
        w.requires_grad or gamma.requires_grad or
        (isinstance(x3, torch.Tensor) and x3.requires_grad))
    if _FUSE_CONV_GN[0] and _PRECISION[0] == 'bf16' and cin >= 16 and grad:
        return ConvGnActFn.apply(x3, w, gamma, beta, groups, eps, stride, pad,
                                 levels, relu, c8_only), out_levels
    if _FUSE_CONV_GN[0] and _PRECISION[0] == 'bf16' and cin >= 16 and not grad:
        # frozen layer (the teacher's towers): the conv output feeds only the
        # norm -- no C8 image of it -- and, ``c8_only``, the layer's output
        # exists only as its C8 image (the next conv's operand)
        raw, _ = conv_forward_raw(x3, w, stride, pad, levels)
        N, c, _ = raw.shape
        lv = levels_desc(out_levels)
        stats = torch.empty((2, N, groups, len(out_levels)),
                            dtype=torch.float32, device=raw.device)
        if c8_only:
            img = _gn_forward_launch(lv, raw, gamma, beta, N, c, groups, eps,
                                     relu, None, stats)
            if img is not None:
                return C8Act(img, raw.shape), out_levels
        y = torch.empty_like(raw)
        _gn_forward_launch(lv, raw, gamma, beta, N, c, groups, eps, relu, y,
                           stats)
        return y, out_levels
    y3, out_levels = conv2d(x3, w, None, stride, pad, levels)
    return gn_act(y3, gamma, beta, groups, eps, out_levels, relu), out_levels


def gn_act(x3, gamma, beta, groups, eps, levels, relu=True):
    if torch.is_grad_enabled() and (x3.requires_grad or gamma.requires_grad):
        return GnActFn.apply(x3, gamma, beta, groups, eps, levels, relu)
    lib = L.get_lib()
    N, c, P = x3.shape
    lv = levels_desc(levels)
    y = torch.empty_like(x3)
    stats = torch.empty((2, N, groups, len(levels)), dtype=torch.float32,
                        device=x3.device)
    _gn_forward_launch(lv, x3, gamma, beta, N, c, groups, eps, relu, y, stats)
    return y


# ---------------------------------------------------------------------------
# small layers
# ---------------------------------------------------------------------------
def copy_into(dst, src):
    """dst.copy_(src) for two contiguous device tensors of one dtype and size, as ONE
    kernel launch of this library (ld_copy_d2d): inside a step that is captured, a
    torch copy is a hipMemcpyAsync, i.e. a 1-D memcpy graph node that a step list
    cannot re-issue (include/ld_hip.h "step lists")."""
    if not (dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous()
            and dst.dtype == src.dtype and dst.numel() == src.numel()):
        dst.copy_(src)
        return dst
    L.check(L.get_lib().ld_copy_d2d(L.ptr(dst), L.ptr(src),
                                    dst.numel() * dst.element_size(),
                                    L.stream_ptr(dst.device)), 'ld_copy_d2d')
    return dst


def maxpool3x3s2(x4):
    """MaxPool2d(3, 2, 1), forward only (frozen stem)."""
    lib = L.get_lib()
    _dev_f32(x4, 'maxpool input')
    if x4.requires_grad and torch.is_grad_enabled():
        raise L.LdError('maxpool3x3s2 has no backward (the stem is frozen: '
                        'frozen_stages >= 0 is required)')
    N, c, h, w = x4.shape
    ho, wo = out_size(h, 3, 2, 1), out_size(w, 3, 2, 1)
    y = torch.empty((N, c, ho, wo), dtype=torch.float32, device=x4.device)
    L.check(lib.ld_maxpool3x3s2(L.ptr(x4), N * c, h, w, L.ptr(y),
                                L.stream_ptr(x4.device)), 'ld_maxpool3x3s2')
    return y


class UpsampleAddFn(torch.autograd.Function):
    """out = fine + nearest_up(coarse)  (FPN top-down step)."""

    @staticmethod
    def forward(ctx, fine, coarse):
        lib = L.get_lib()
        _dev_f32(fine, 'fine')
        _dev_f32(coarse, 'coarse')
        N, c, hf, wf = fine.shape
        hc, wc = coarse.shape[2:]
        out = torch.empty_like(fine)
        L.check(lib.ld_upsample_add_forward(L.ptr(fine), L.ptr(coarse), N * c,
                                            hf, wf, hc, wc, L.ptr(out),
                                            L.stream_ptr(fine.device)),
                'ld_upsample_add_forward')
        ctx.shapes = (N, c, hf, wf, hc, wc)
        ctx.fan_f, ctx.fan_c = fan_in(ctx, 0, fine), fan_in(ctx, 1, coarse)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.get_lib()
        N, c, hf, wf, hc, wc = ctx.shapes
        dout = dout.contiguous()
        dcoarse = None
        if ctx.needs_input_grad[1]:
            dcoarse = torch.empty((N, c, hc, wc), dtype=torch.float32,
                                  device=dout.device)
            # the coarse map's other consumer (its 3x3 output conv) ran first
            # and deposited its gradient: summed here
            addend = fan_take(ctx.fan_c)
            L.check(lib.ld_upsample_add_backward_acc(
                L.ptr(dout), N * c, hf, wf, hc, wc, L.ptr(addend),
                L.ptr(dcoarse), L.stream_ptr(dout.device)),
                'ld_upsample_add_backward_acc')
        dfine = dout if ctx.needs_input_grad[0] else None
        return fan_give(ctx.fan_f, dfine), fan_give(ctx.fan_c, dcoarse)


def upsample_add(fine, coarse):
    return UpsampleAddFn.apply(fine, coarse)


class ScaleLevelsFn(torch.autograd.Function):
    """y[:, :, level l] = x * scales[l]  (mmcv Scale per FPN level)."""

    @staticmethod
    def forward(ctx, x3, scales, levels):
        lib = L.get_lib()
        _dev_f32(x3, 'scale input')
        N, c, P = x3.shape
        lv = levels_desc(levels)
        y = torch.empty_like(x3)
        L.check(lib.ld_scale_levels_forward(C.byref(lv), L.ptr(x3),
                                            L.ptr(scales), N * c, L.ptr(y),
                                            L.stream_ptr(x3.device)),
                'ld_scale_levels_forward')
        ctx.save_for_backward(x3, scales)
        ctx.levels = levels
        ctx.params = (scales, )
        _note_use(scales)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.get_lib()
        x3, scales = ctx.saved_tensors
        dy = dy.contiguous()
        N, c, P = x3.shape
        lv = levels_desc(ctx.levels)
        dx = torch.empty_like(x3)
        ps, = ctx.params
        sink = _sink(ps) if ctx.needs_input_grad[1] else None
        if sink is not None:
            ds = sink
        else:
            ds = torch.empty_like(scales) if ctx.needs_input_grad[1] else None
        need = lib.ld_scale_levels_backward_workspace_bytes(C.byref(lv))
        ws = workspace(x3.device, need, 'scale_bwd')
        L.check(lib.ld_scale_levels_backward(
            C.byref(lv), L.ptr(dy), L.ptr(x3), L.ptr(scales), N * c, L.ptr(dx),
            L.ptr(ds), 0 if sink is None else 1, L.ptr(ws), ws.numel(),
            L.stream_ptr(x3.device)), 'ld_scale_levels_backward')
        if sink is not None:
            ds = None
            _emit(ps)
        return dx, ds, None


def scale_levels(x3, scales, levels):
    return ScaleLevelsFn.apply(x3, scales, levels)


def deform_im2col(x3, off3, h, w, k, stride, pad, dilation):
    """(N, Cin, H*W) + offsets (N, 2*k*k, Hout*Wout) -> column tensor
    (N, Cin*k*k, Hout*Wout), forward only (ld_deform_im2col)."""
    lib = L.get_lib()
    _dev_f32(x3, 'dcn input')
    _dev_f32(off3, 'dcn offset')
    N, cin, P = x3.shape
    ho, wo = out_size(h, k, stride, pad), out_size(w, k, stride, pad)
    if P != h * w or off3.shape != (N, 2 * k * k, ho * wo):
        raise L.LdError('deform_im2col: shape mismatch')
    col = torch.empty((N, cin * k * k, ho * wo), dtype=torch.float32,
                      device=x3.device)
    L.check(lib.ld_deform_im2col(L.ptr(x3), L.ptr(off3), N, cin, h, w, k, k,
                                 stride, pad, dilation, L.ptr(col),
                                 L.stream_ptr(x3.device)), 'ld_deform_im2col')
    return col


def gconv_weight_image(w, groups):
    """[group][ci][tap][co] image of a grouped conv weight (Cout, Cin/groups, K,
    K), cached on the tensor (the X-101 teacher is frozen)."""
    lib = L.get_lib()
    _dev_f32(w, 'grouped conv weight')
    stamp = (w._version, w.data_ptr(), groups)
    hit = getattr(w, '_ld_gimg', None)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    cout, cin_g, k, _ = w.shape
    img = torch.empty(lib.ld_gconv_weight_image_floats(cout, cin_g * groups,
                                                       groups, k),
                      dtype=torch.float32, device=w.device)
    L.check(lib.ld_gconv_weight_transform(
        L.ptr(w.detach().contiguous()), cout, cin_g * groups, groups, k,
        L.ptr(img), L.stream_ptr(w.device)), 'ld_gconv_weight_transform')
    try:
        w._ld_gimg = (stamp, img)
    except AttributeError:
        pass
    return img


def gconv_forward(x3, w, groups, stride, pad, levels, scale=None, shift=None,
                  relu=False):
    """Grouped conv forward (ld_gconv_forward), single level, forward only:
    (N, Cin, H*W) -> ((N, Cout, Ho*Wo), ((Ho, Wo),))."""
    lib = L.get_lib()
    _dev_f32(x3, 'grouped conv input')
    if len(levels) != 1:
        raise NotImplementedError('grouped conv on level-concatenated tensors')
    (h, wd), = levels
    N, cin, P = x3.shape
    cout, cin_g, k, _ = w.shape
    if P != h * wd or cin != cin_g * groups:
        raise L.LdError('gconv_forward: shape mismatch')
    ho, wo = out_size(h, k, stride, pad), out_size(wd, k, stride, pad)
    img = gconv_weight_image(w, groups)
    y = torch.empty((N, cout, ho * wo), dtype=torch.float32, device=x3.device)
    L.check(lib.ld_gconv_forward(
        L.ptr(x3.contiguous()), L.ptr(img), L.ptr(y), N, cin, cout, groups, k,
        stride, pad, h, wd, L.ptr(scale), L.ptr(shift), 1 if relu else 0,
        L.stream_ptr(x3.device)), 'ld_gconv_forward')
    return y, ((ho, wo), )


class QualityFn(torch.autograd.Function):
    """GFLv2's distribution-guided quality branch on level-concatenated
    tensors (gfocal_head.py:201-217): (reg3 (N, 68, P), cls_feat3 (N, C, P),
    reg_conf parameters) -> cls_score3 = sigmoid(cls_feat3) * quality."""

    @staticmethod
    def forward(ctx, reg3, cls_feat3, w1, b1, w2, b2):
        lib = L.get_lib()
        for t, n in ((reg3, 'reg'), (cls_feat3, 'cls_feat'), (w1, 'w1'),
                     (b1, 'b1'), (w2, 'w2'), (b2, 'b2')):
            _dev_f32(t, 'quality ' + n)
        N, c, P = cls_feat3.shape
        if reg3.shape != (N, 68, P) or w1.shape[:2] != (64, 20) or \
                w2.numel() != 64:
            raise L.LdError('quality branch: reg_max=16, reg_topk=4, add_mean, '
                            'reg_channels=64 are compiled in')
        cls_score = torch.empty_like(cls_feat3)
        quality = torch.empty((N, P), dtype=torch.float32, device=reg3.device)
        L.check(lib.ld_quality_forward(
            L.ptr(reg3), L.ptr(cls_feat3), N, c, P, L.ptr(w1), L.ptr(b1),
            L.ptr(w2), L.ptr(b2), L.ptr(cls_score), L.ptr(quality),
            L.stream_ptr(reg3.device)), 'ld_quality_forward')
        ctx.save_for_backward(reg3, cls_feat3, quality, w1, b1, w2, b2)
        ctx.params = (w1, b1, w2, b2)
        _note_use(w1, b1, w2, b2)
        ctx.mark_non_differentiable(quality)
        return cls_score, quality

    @staticmethod
    def backward(ctx, g_cls_score, _g_quality):
        lib = L.get_lib()
        reg3, cls_feat3, quality, w1, b1, w2, b2 = ctx.saved_tensors
        N, c, P = cls_feat3.shape
        g = g_cls_score.contiguous()
        g_cf = torch.empty_like(cls_feat3)
        g_reg = torch.empty_like(reg3)
        sinks = [_sink(p) for p in ctx.params]
        direct = all(s is not None for s in sinks)
        if direct:
            gp = sinks
        else:
            gp = [torch.empty_like(p) for p in ctx.params]
        need = lib.ld_quality_backward_workspace_bytes(N, P)
        ws = workspace(reg3.device, need, 'quality_bwd')
        L.check(lib.ld_quality_backward(
            L.ptr(reg3), L.ptr(cls_feat3), L.ptr(quality), L.ptr(g), N, c, P,
            L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(g_cf),
            L.ptr(g_reg), L.ptr(gp[0]), L.ptr(gp[1]), L.ptr(gp[2]),
            L.ptr(gp[3]), 1 if direct else 0, L.ptr(ws), ws.numel(),
            L.stream_ptr(reg3.device)), 'ld_quality_backward')
        if direct:
            for p in ctx.params:
                _emit(p)
            gp = [None] * 4
        return (g_reg, g_cf) + tuple(gp)


def sgd_step(params_flat, grads_flat, momentum_flat, lr, momentum,
             weight_decay, grad_scale=1.0, hyper=None):
    """``hyper``: optional device tensor [lr, momentum, weight_decay,
    grad_scale]; when given the kernel reads the four values from it (the
    by-value arguments are ignored) -- the form a captured hipGraph needs."""
    lib = L.get_lib()
    for t in (params_flat, grads_flat, momentum_flat):
        _dev_f32(t, 'sgd arena')
    if hyper is not None:
        _dev_f32(hyper, 'sgd hyper-parameters')
        L.check(lib.ld_sgd_step_dev(
            L.ptr(params_flat), L.ptr(grads_flat), L.ptr(momentum_flat),
            params_flat.numel(), L.ptr(hyper),
            L.stream_ptr(params_flat.device)), 'ld_sgd_step_dev')
    else:
        L.check(lib.ld_sgd_step(L.ptr(params_flat), L.ptr(grads_flat),
                                L.ptr(momentum_flat), params_flat.numel(), lr,
                                momentum, weight_decay, grad_scale,
                                L.stream_ptr(params_flat.device)),
                'ld_sgd_step')
    bump_param_generation()
    refresh_params(params_flat.device)


# ---------------------------------------------------------------------------
# level packing: the shared head towers run on ONE level-concatenated tensor
# ---------------------------------------------------------------------------
def _level_ptrs(ts):
    L.keep(*ts)
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _pack_launch(feats, levels):
    f0 = feats[0]
    N, c = int(f0.shape[0]), int(f0.shape[1])
    P = sum(h * w for h, w in levels)
    x3 = torch.empty((N, c, P), dtype=torch.float32, device=f0.device)
    fs = [_dev_f32(f, 'level map') for f in feats]
    lv = levels_desc(levels)
    if _C8[0] and _PRECISION[0] == 'bf16' and c % 32 == 0:
        # bf16 mode: the C8 image of the packed tensor from the same launch (the
        # first tower convs read it)
        x8 = torch.empty(N * c * P, dtype=torch.bfloat16, device=f0.device)
        L.check(L.get_lib().ld_pack_levels_c8(
            C.byref(lv), _level_ptrs(fs), N, c, L.ptr(x3), L.ptr(x8),
            L.stream_ptr(f0.device)), 'ld_pack_levels_c8')
        _attach_c8(x3, x8)
        return x3
    L.check(L.get_lib().ld_pack_levels(C.byref(lv), _level_ptrs(fs), N * c,
                                       L.ptr(x3), L.stream_ptr(f0.device)),
            'ld_pack_levels')
    return x3


class PackLevelsFn(torch.autograd.Function):
    """tuple of (N, C, H_l, W_l) -> (N, C, P): one launch each way (torch.cat
    forward and five strided-slice copies backward before)."""

    @staticmethod
    def forward(ctx, levels, *feats):
        ctx.levels = levels
        ctx.fans = [fan_in(ctx, 1 + i, f) for i, f in enumerate(feats)]
        ctx.shape = tuple(feats[0].shape[:2])
        return _pack_launch(feats, levels)

    @staticmethod
    def backward(ctx, dx3):
        N, c = ctx.shape
        dx3 = dx3.contiguous()
        outs = [torch.empty((N, c, h, w), dtype=torch.float32,
                            device=dx3.device) for h, w in ctx.levels]
        lv = levels_desc(ctx.levels)
        if _C8[0] and _PRECISION[0] == 'bf16' and c % 32 == 0:
            # the level gradients feed the neck convs' C8 data / weight gradients
            o8 = [torch.empty(N * c * h * w, dtype=torch.bfloat16,
                              device=dx3.device) for h, w in ctx.levels]
            L.check(L.get_lib().ld_unpack_levels_c8(
                C.byref(lv), L.ptr(dx3), N, c, _level_ptrs(outs),
                _level_ptrs(o8), L.stream_ptr(dx3.device)),
                'ld_unpack_levels_c8')
            for o, b in zip(outs, o8):
                _attach_c8(o, b)
        else:
            L.check(L.get_lib().ld_unpack_levels(
                C.byref(lv), L.ptr(dx3), N * c, _level_ptrs(outs),
                L.stream_ptr(dx3.device)), 'ld_unpack_levels')
        return (None, ) + tuple(fan_give(f, o)
                                for f, o in zip(ctx.fans, outs))


def pack_levels(feats):
    """tuple of (N, C, H_l, W_l) -> (N, C, P) level-concatenated tensor."""
    levels = tuple((int(f.shape[2]), int(f.shape[3])) for f in feats)
    if len(feats) == 1:
        return feats[0].reshape(feats[0].shape[0], feats[0].shape[1], -1), levels
    if any(not f.is_contiguous() for f in feats):
        return torch.cat([f.flatten(2) for f in feats], dim=2), levels
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        return PackLevelsFn.apply(levels, *feats), levels
    return _pack_launch(feats, levels), levels


def _common_buffer(gs, levels, N, c):
    """The (N, C, P) tensor the per-level gradients ``gs`` are views of, or
    None."""
    P = sum(h * w for h, w in levels)
    g0 = gs[0]
    if g0 is None or g0.dtype != torch.float32:
        return None
    st0, off = g0.untyped_storage().data_ptr(), 0
    base = g0.storage_offset()
    for g, (h, w) in zip(gs, levels):
        if g is None or tuple(g.shape) != (N, c, h, w) or \
                g.untyped_storage().data_ptr() != st0 or \
                g.stride() != (c * P, P, w, 1) or \
                g.storage_offset() != base + off:
            return None
        off += h * w
    return torch.as_strided(g0, (N, c, P), (c * P, P, 1), base)


class SplitLevelsFn(torch.autograd.Function):
    """(N, C, P) -> per-level (N, C, H_l, W_l) VIEWS.  Backward: when the
    incoming gradients are the level views of one (N, C, P) buffer (the loss
    block writes them that way) that buffer IS the gradient -- autograd's own
    slice backward allocated a zero-filled full-size tensor per level and summed
    the five (10 fills + 10 copies + 8 adds of 12-14 MB per step)."""

    @staticmethod
    def forward(ctx, x3, levels):
        ctx.levels, ctx.shape = levels, tuple(x3.shape)
        ctx.fan = fan_in(ctx, 0, x3)
        return tuple(level_views(x3, levels))

    @staticmethod
    def backward(ctx, *gs):
        N, c, P = ctx.shape
        buf = _common_buffer(gs, ctx.levels, N, c)
        if buf is None:
            ref = next(g for g in gs if g is not None)
            full = [g.contiguous() if g is not None else
                    torch.zeros((N, c, h, w), dtype=torch.float32,
                                device=ref.device)
                    for g, (h, w) in zip(gs, ctx.levels)]
            buf = _pack_launch(full, ctx.levels)
        return fan_give(ctx.fan, buf), None


def split_levels(x3, levels):
    """(N, C, P) -> list of (N, C, H_l, W_l) views (no copies)."""
    if torch.is_grad_enabled() and x3.requires_grad and x3.is_contiguous():
        return list(SplitLevelsFn.apply(x3, tuple(levels)))
    return level_views(x3, levels)
