#!/usr/bin/env python
"""bench.py -- images/sec of the LD train step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

With N > 1 and no launcher in the environment bench.py starts the N ranks
itself (torch.distributed.run, one rank per GPU over RCCL); it exits non-zero
when the node has fewer than N GPUs or the launcher's WORLD_SIZE is not N.

One step = student (GFocal-R50) + frozen teacher (GFocal-R101) dual forward,
batched ATSS/VLR/IM targets, fused LD loss block, backward, bucketed RCCL
gradient all-reduce, fused SGD -- on a synthetic COCO-shape batch (2 images
per GPU, 800x1333 padded to 800x1344, 7 GT boxes each) resident in HBM.
Workload = BASELINE.json configs[1] (ld_r50_gflv1_r101_fpn_coco_1x, bs 2 per
GPU, fp32); weak scaling (per-GPU batch fixed).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline      the dominant kernel family (fp32 MFMA implicit-GEMM conv):
                algorithmic conv FLOPs / summed HIP-event launch durations,
                measured in an instrumented pass right after the timed region
  roofline_ldkl the north-star fused LD-KL + Integral kernel at a saturating
                2^24-row size against the HBM roofline
  cpu_baseline  the reference's own train step (kind "reference": unmodified
                mmdet sources from the oracle/_ref archive, through the oracle
                shim, in a child process) on the host cores; the CPU oracle
                ("port") only when that archive is absent.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# The step runs on three HIP streams (student, teacher one step ahead, weight
# gradients) and RCCL adds its own.  The ROCm runtime multiplexes streams onto 4
# hardware queues by default: once a process group exists two of the step's
# streams land on ONE queue and the teacher overlap is gone (measured with the
# collectives forced in a 1-rank group: 36.8 ms per step at the default, 35.1-35.5
# with 5 / 6 / 8 queues = the figure without a process group).  hipGraph replays
# need DEBUG_HIP_FORCE_GRAPH_QUEUES=2 next to it (the graph executor's internal
# streams must not each get a hardware queue of their own: bf16 replay 28 ms vs
# 15.2, profiles/r05_graph_queues_s1.jsonl; ld_amd/__init__.py, DESIGN.md section
# 6).  The runtime reads both when it initialises, at the first HIP call.
if int(os.environ.get('WORLD_SIZE', '1')) > 1 or \
        os.environ.get('LD_FORCE_COLLECTIVES') == '1':
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    os.environ.setdefault('DEBUG_HIP_FORCE_GRAPH_QUEUES', '2')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = 'images/sec (1333x800) GFocal-R50<-R101 LD train step'
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-per-gpu', type=int, default=2)
    ap.add_argument('--num-gt', type=int, default=7)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-roofline', action='store_true')
    ap.add_argument('--profile-steps', type=int, default=2)
    ap.add_argument('--no-graph', action='store_true',
                    help='skip the step-list leg (the captured step re-issued by '
                    'the C launch loop)')
    ap.add_argument('--no-bf16', action='store_true',
                    help='skip the bf16 (BASELINE config 3) leg')
    ap.add_argument('--no-prefetch', action='store_true',
                    help='run the frozen teacher inside each step instead of '
                    'one step ahead (see "teacher_prefetch" in the output)')
    ap.add_argument('--config', type=int, default=2, choices=(2, 4, 5),
                    help='BASELINE.json configs[] entry (1-based): 2 = the '
                    'headline R50<-R101 fp32 step (default); 4 = R101 <- '
                    'R101-DCN (ld_r101_gflv1_r101dcn_fpn_coco_2x); 5 = LDv2 '
                    'R50 <- X101 \'finegrained\'')
    return ap.parse_args()


# per-config workload: detector builder, name, analytic conv GFLOP per image
# 
def _workload(cfg_id):
    from ld_amd import model_zoo
    if cfg_id == 4:
        return (lambda dev: model_zoo.build_seeded(
            model_zoo.ld_r101_dcn_detector(), dev),
            'ld_r101_gflv1_r101dcn_fpn_coco_2x (BASELINE.json configs[3]): '
            'GFocal-R101 student <- R101-DCN(c3-c5) teacher, fp32', 2324.0)
    if cfg_id == 5:
        return (lambda dev: model_zoo.build_seeded(
            model_zoo.ldv2_x101_detector(), dev),
            'ldv2 R50 <- X101 (32x4d), imitation \'finegrained\' (BASELINE.json '
            'configs[4], the composition of SURVEY Q10): LDv2Head student <- '
            'GFLv2 ResNeXt-101 teacher, fp32', None)
    return (lambda dev: model_zoo.build_seeded_ld_detector(50, 101, dev),
            'ld_r50_gflv1_r101_fpn_coco_1x (BASELINE.json configs[1]): '
            'GFocal-R50 student <- R101 teacher, fp32', 1835.7)


def make_batch(bs, num_gt, seed, dev):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(bs, (800, 1333), (800, 1344), num_gt, seed)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    return b, d


def kernel_roofline(trainer, dbatch, steps, bf16=False):
    """Instrumented pass: HIP events around every conv launch, teacher on the
    main stream so durations do not overlap."""
    from ld_amd import layers as Y
    model = trainer.model
    prev = getattr(model, 'use_teacher_stream', False)
    model.use_teacher_stream = False
    trainer.step(dbatch)
    torch.cuda.synchronize()
    with Y.KernelProfile() as prof:
        for _ in range(steps):
            trainer.step(dbatch)
    agg = prof.summary()
    alg_bytes = prof.algorithmic_bytes()
    alg_rd, alg_wr = prof.algorithmic_read_write()
    fus_rd, fus_wr = prof.fused_read_write()
    model.use_teacher_stream = prev
    by_kind = {k: dict(ms_per_step=v[0] / steps * 1e3,
                       tflops=v[1] / v[0] / 1e12 if v[0] else 0.0,
                       launches=v[2] / steps) for k, v in agg.items()}
    if bf16:
        main = {k: v for k, v in agg.items() if k.endswith('_bf16')}
        rest = {k: v for k, v in agg.items() if not k.endswith('_bf16')}
        peak = PEAK_BF16_MFMA_TFLOPS
        kernel = ('conv (bf16 MFMA 32x32x16, fp32 accumulate: LDS-tiled / '
                  'streaming fwd+dgrad and transpose-read wgrad on bf16 C8 '
                  'operand images, conv_bf16.hip)')
    else:
        main, rest, peak = agg, {}, PEAK_FP32_MFMA_TFLOPS
        kernel = ('conv (fp32 MFMA 32x32x2 implicit GEMM: streaming fwd/dgrad, '
                  'workgroup-tiled / three-taps-per-workgroup wgrad with '
                  'in-workgroup k-groups, conv.hip + conv_wgrad.hip)')
    tot_t = sum(v[0] for v in main.values())
    tot_f = sum(v[1] for v in main.values())
    tot_n = sum(v[2] for v in main.values())
    ach = tot_f / tot_t / 1e12 if tot_t > 0 else 0.0
    alg_per_launch = alg_bytes / max(tot_n, 1)
    # `traffic` is per conv call of THIS line (the PMC total per step divided by
    # this line's own launches_per_step), so that traffic / algorithmic_bytes_per_
    # launch is the per-step ratio: ONE ratio, in the fields and in the note
    pmc = PMC_CONV['bf16' if bf16 else 'fp32']
    pmc_per_launch = (pmc['fetch'] + pmc['write']) / max(tot_n / steps, 1)
    ratio = pmc_per_launch / alg_per_launch if alg_per_launch else 0.0
    rd_ratio = pmc['fetch'] / (alg_rd / steps) if alg_rd else 0.0
    wr_ratio = pmc['write'] / (alg_wr / steps) if alg_wr else 0.0
    out = dict(
        kernel=kernel, bound='mfma', achieved=ach, peak=peak,
        unit='TFLOP/s', frac=ach / peak,
        traffic=pmc_per_launch,
        algorithmic_bytes_per_launch=alg_per_launch,
        traffic_over_algorithmic=ratio,
        algorithmic_read_bytes_per_launch=alg_rd / max(tot_n, 1),
        algorithmic_write_bytes_per_launch=alg_wr / max(tot_n, 1),
        fetch_over_algorithmic_reads=rd_ratio,
        write_over_algorithmic_writes=wr_ratio,
        # what the fused epilogues read (residual, gradient addend) and write
        # (the raw second output of conv+BN launches) on top, by design; the
        # ratios with those bytes in the denominator
        fused_epilogue_read_bytes_per_step=fus_rd / steps,
        fused_epilogue_write_bytes_per_step=fus_wr / steps,
        fetch_over_reads_incl_fused=None if not alg_rd else
        pmc['fetch'] / ((alg_rd + fus_rd) / steps),
        write_over_writes_incl_fused=None if not alg_wr else
        pmc['write'] / ((alg_wr + fus_wr) / steps),
        traffic_note=(
            'fabric-side bytes of the GEMM kernels per step (requests leaving the '
            'XCD L2s, Infinity-Cache hits included) divided by this line\'s '
            'launches_per_step: rocprofv3 --pmc FETCH_SIZE (x 2: calibrated on a '
            'known-size copy at 4 B and 16 B per lane, '
            'profiles/r04_pmc_calib_copy_*) and WRITE_SIZE in separate passes of '
            'tools/profile_step.py --serial (%d dispatches per step), %s = %.2f x '
            'algorithmic_bytes_per_launch (both operands + the output once, the '
            'field next to it); by direction: reads %.2f x, writes %.2f x.  The '
            'algorithmic figure leaves out what the fused epilogues read and '
            'write on purpose (the residual / gradient addend, the second output '
            'of conv+BN launches) and counts the weights once where each of the '
            '8 XCD L2s fetches its own copy; the kernels are MFMA-bound, but the '
            'bytes through the L1 miss path are what the time above the MFMA '
            'floor is made of (profiles/r04_wgrad_attribution.txt); not '
            're-measured inside bench.py') % (
                pmc['dispatches'], pmc['file'], ratio, rd_ratio, wr_ratio),
        launches_per_step=tot_n / steps,
        avg_launch_us=tot_t / max(tot_n, 1) * 1e6,
        conv_ms_per_step=tot_t / steps * 1e3,
        gflop_per_step=tot_f / steps / 1e9,
        by_kind=by_kind)
    if rest:
        out['fp32_fallback_conv_ms_per_step'] = \
            sum(v[0] for v in rest.values()) / steps * 1e3
    return out


def _reg_dense_launcher(dev, sizes, strides, n_img, density, seed=0):
    """A callable that enqueues the train step's reg-side dense kernel
    (loss_reg_dense_kernel through ld_loss_main_parts(LD_LOSS_PART_REG)) on
    synthetic maps of the given pyramid; ``density`` = fraction of anchors in
    the valuable-localisation region (weight > 0), no positives."""
    import ctypes as C
    from ld_amd import lib as L
    from ld_amd import lossblock as LB
    lib = L.get_lib()
    g = torch.Generator(device='cpu').manual_seed(seed)
    geom = L.make_geom(sizes, strides, n_img)
    A = geom.num_anchors
    s_reg = [torch.randn(n_img, 68, h, w, device=dev) * 3 for h, w in sizes]
    t_reg = [torch.randn(n_img, 68, h, w, device=dev) * 3 for h, w in sizes]
    g_reg = [torch.empty_like(t) for t in s_reg]
    labels = torch.full((n_img, A), 80, dtype=torch.int64, device=dev)
    lw = torch.ones((n_img, A), device=dev)
    bt = torch.zeros((n_img, A, 4), device=dev)
    vlr = torch.rand((n_img, A), device=dev) + 1e-3
    if density < 1.0:
        vlr = vlr * (torch.rand((n_img, A), device=dev) < density)
    zeros = torch.zeros((n_img, A), device=dev)
    counts = torch.zeros(n_img + 2 * len(sizes) + 1, dtype=torch.int32,
                         device=dev)
    norm = torch.ones(4, device=dev)
    hp = LB.make_hp()
    ws = LB.workspace(dev, lib.ld_loss_workspace_bytes(C.byref(geom)),
                      'bench_loss')
    m_s, m_t, m_g = L.make_maps(s_reg), L.make_maps(t_reg), L.make_maps(g_reg)
    st = L.stream_ptr(dev)
    keep = (s_reg, t_reg, g_reg, labels, lw, bt, vlr, zeros, counts, norm, ws)

    def launch():
        L.check(lib.ld_loss_main_parts(
            C.byref(geom), C.byref(hp), C.byref(m_s), C.byref(m_s),
            C.byref(m_t), C.byref(m_t), C.byref(m_s), C.byref(m_t),
            L.ptr(labels), L.ptr(lw), L.ptr(bt), L.ptr(vlr), L.ptr(zeros),
            L.ptr(counts), L.ptr(zeros), L.ptr(zeros), L.ptr(norm), None,
            C.byref(m_g), C.byref(m_g), C.byref(m_g), None, None, None,
            L.ptr(ws), ws.numel(), 2, st), 'ld_loss_main_parts')

    launch.keep = keep
    return launch, n_img * A * 4


def _median_launch_us(launch, warm, iters):
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True),
            torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        launch()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2], ts[0]


# profiles/r03_pmc_traffic_regdense.txt (the round-3 kernel): FETCH_SIZE
# 1 212 475.4 KB (x 2, gfx950 correction) + WRITE_SIZE 1 115 649.4 KB per launch
# at 2^24 rows
PMC_LDKL_TRAFFIC_BYTES = (2 * 1212475.4 + 1115649.4) * 1024.0
# profiles/r0N_pmc_traffic_conv_step_*_by_kernel.txt (the serialised
# fp32 step, 6 steps): the GEMM kernels (forward / dgrad streaming shapes + the
# weight-gradient kernels; 324 dispatches per step incl. the stride-2 dgrad parity
# classes, 316 conv calls) move FETCH_SIZE 2 x 12.12 GB + WRITE_SIZE 9.62 GB =
# 33.87 GB per step.  FETCH_SIZE x 2 is CALIBRATED for this access width: a 1 GiB
# copy reports exactly half its read bytes at 4 B per lane and at 16 B per lane
# (profiles/r04_pmc_calib_copy_*): the counter tallies 128-byte fabric requests at
# 64 bytes.  It counts requests LEAVING an XCD's L2, Infinity-Cache hits included.
# (The per-bucket slab reduce adds 1.7 GB per step; not part of `traffic`.)
# The constants come out of tools/pmc_conv_bytes.py <file> <steps in the run>, on the
# round-6 final build (tools/sessions/r6_pmc.sh; the serialised step, FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes):
#   fp32: 6 steps, 324 GEMM dispatches per step: FETCH 2 x 12.13 + WRITE 9.63 GB
#   bf16: 7 steps, 282 GEMM dispatches per step (C8 tile / LDS-DMA / C8 weight-
#         gradient kernels, the fused teacher bottleneck): FETCH 2 x 5.51 + WRITE 9.08 GB
PMC_CONV = {
    'fp32': dict(file='profiles/r06_pmc_traffic_conv_step_fp32_by_kernel.txt',
                 dispatches=324, fetch=24.261e9, write=9.625e9),
    'bf16': dict(file='profiles/r06_pmc_traffic_conv_step_bf16_by_kernel.txt',
                 dispatches=282, fetch=10.691e9, write=7.453e9),
}


def hbm_ceilings(dev):
    """What the memory system delivers, measured in THIS process right next to
    the LD-KL kernel: a 16-byte-per-lane
    non-temporal copy, and the kernel's own 34-read / 17-write channel-plane
    pattern with the arithmetic removed (ld_probe_copy / ld_probe_planes,
    ld_amd/csrc/probe.hip)."""
    import ctypes as C
    from ld_amd import lib as L
    lib = L.get_lib()
    st = L.stream_ptr(dev)
    out = {}
    n = 1 << 28  # 1 GiB per array: well past the 256 MiB Infinity Cache
    src = torch.randn(n, device=dev)
    dst = torch.empty_like(src)

    def copy():
        L.check(lib.ld_probe_copy(L.ptr(src), L.ptr(dst), n, 4, 1, st),
                'ld_probe_copy')
    us, _ = _median_launch_us(copy, 3, 11)
    out['copy_16B_nt_GBps'] = 8.0 * n / (us * 1e-6) / 1e9
    del src, dst
    rows = 1 << 22
    s = torch.randn(68 * rows, device=dev)
    t = torch.randn(68 * rows, device=dev)
    g = torch.empty_like(s)

    def planes():
        L.check(lib.ld_probe_planes(L.ptr(s), L.ptr(t), L.ptr(g), rows, 1, 1,
                                    st), 'ld_probe_planes')
    us, _ = _median_launch_us(planes, 3, 11)
    out['pattern_no_math_GBps'] = 204.0 * 4 * rows / (us * 1e-6) / 1e9
    out['note'] = ('copy = float4 non-temporal copy of 1 GiB; pattern = the '
                   'kernel\'s 34 read + 17 write channel planes per side '
                   '(2^22 x 4 row-sides, side-fast mapping) with out = s - t '
                   'instead of the KL')
    del s, t, g
    torch.cuda.empty_cache()
    return out


def ldkl_roofline(dev):
    """North-star kernel = the reg-side dense kernel THE TRAIN STEP LAUNCHES
    (fused LD-KL + VLR-LD + Integral chain, forward + gradient), timed with HIP
    events at a saturating size (2^24 anchor-side rows, every anchor in the
    VLR region) against the HBM roofline, and at the C2 step size (launch
    bound: absolute microseconds only).  Algorithmic bytes per anchor-side row:
    136 (17 student + 17 teacher logits) + 68 (gradient) + 3 (label int64 +
    VLR weight per anchor, / 4 sides) = 207."""
    bytes_per_row = 207.0
    launch, rows = _reg_dense_launcher(dev, [(2048, 2048)], [8], 1, 1.0)
    us, us_min = _median_launch_us(launch, 10, 31)
    ach = rows * bytes_per_row / (us * 1e-6) / 1e9
    sparse, _ = _reg_dense_launcher(dev, [(2048, 2048)], [8], 1, 0.09, seed=1)
    us_sparse, _ = _median_launch_us(sparse, 5, 15)
    del launch, sparse
    c2, rows_c2 = _reg_dense_launcher(
        dev, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)],
        [8, 16, 32, 64, 128], 2, 0.09, seed=2)
    us_c2, _ = _median_launch_us(c2, 10, 51)
    del c2
    ceil = hbm_ceilings(dev)
    return dict(kernel='loss_reg_lean_kernel (the train step\'s fused LD-KL + '
                       'VLR-LD + Integral chain, fwd+grad) via '
                       'ld_loss_main_parts(LD_LOSS_PART_REG)',
                ceilings_same_process=ceil,
                frac_of_copy_ceiling=ach / ceil['copy_16B_nt_GBps'],
                frac_of_pattern_ceiling=ach / ceil['pattern_no_math_GBps'],
                bound='hbm', achieved=ach, peak=PEAK_HBM_GBPS, unit='GB/s',
                frac=ach / PEAK_HBM_GBPS,
                traffic=PMC_LDKL_TRAFFIC_BYTES if rows == 1 << 24 else None,
                traffic_note='HBM bytes per launch of THIS kernel at THIS size '
                             'from separate rocprofv3 --pmc passes (FETCH_SIZE '
                             'x 2 per MI355X_MICROARCH.md + WRITE_SIZE), '
                             'profiles/r03_pmc_traffic_regdense.txt '
                             '(tools/pmc_traffic.sh + tools/one_regdense.py): '
                             '1.04 x the algorithmic bytes; not re-measured '
                             'inside bench.py',
                rows=rows, bytes_per_row=bytes_per_row, us=us, us_min=us_min,
                us_vlr_density_0p09=us_sparse,
                c2_rows=rows_c2, c2_us=us_c2)


def _host_cpu():
    model, phys = 'unknown', None
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        pass
    return model, phys, os.cpu_count() or 1


def cpu_baseline_reference(threads, reps=3, timeout_s=600):
    """The REFERENCE'S OWN train step on the host cores:
    KnowledgeDistillationSingleStageDetector.forward_train -> _parse_losses ->
    backward -> SGD.step, imported unmodified from the archive
    oracle/_ref/reference_snapshot.tar.gz (packed from /root/reference by
    oracle/make_ref_snapshot.py at build time; it travels with the gpurun
    snapshot like the built .so) through oracle/ref_shim.py, in a CHILD process
    (oracle/ref_cpu_step.py).  Returns None when the archive is absent."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import make_ref_snapshot as MRS
    if not os.path.exists(MRS.ARCHIVE):
        return None
    root = tempfile.mkdtemp(prefix='ld_ref_')
    try:
        MRS.extract(root)
        env = dict(os.environ, LD_REFERENCE_ROOT=root, OMP_NUM_THREADS=str(threads))
        r = subprocess.run(
            [sys.executable, os.path.join(REPO, 'oracle', 'ref_cpu_step.py'),
             '--threads', str(threads), '--reps', str(reps)],
            env=env, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            print(f'[bench] reference CPU step failed: {r.stderr[-800:]}',
                  file=sys.stderr)
            return None
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f'[bench] reference CPU step failed: {e!r}', file=sys.stderr)
        return None
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cpu_baseline(batch, sdepth=50, tdepth=101, reps=3):
    """BASELINE.md section 3 on the GPU box's host cores, on the SAME synthetic
    batch shape and seeded weights as the device run.  kind "reference": the
    reference's own code (cpu_baseline_reference above), 1 warm-up + the median
    of 3 steps, per-stage split.  When the archive is absent (a tree that was
    never built next to /root/reference) the CPU oracle "port" (torch-CPU fp32
    nets with the reference's layer sequence + the numpy loss block) is timed
    instead and labelled so; the port was measured at 0.82 x the reference's
    rate in the build container (profiles/r02_cpu_reference_baseline.json)."""
    model, phys, logical = _host_cpu()
    threads = min(phys or logical, 64)  # oneDNN stops scaling past one socket
    n = batch['img'].shape[0]
    ref = cpu_baseline_reference(threads)
    if ref is not None:
        st = ref['stages_s']
        return dict(value=ref['images'] / st['total'], unit='images/sec',
                    cores=threads, kind='reference', cpu_model=model,
                    physical_cores=phys, logical_cpus=logical,
                    stages_s={k: round(v, 3) for k, v in st.items() if k != 'loss'},
                    loss_block_s=round(st.get('loss_block', 0.0), 4),
                    last_loss=st.get('loss'),
                    all_totals_s=ref['all_totals_s'],
                    sample=f'the reference\'s own LD train step (kd_one_stage.'
                    f'forward_train + _parse_losses + backward + SGD.step, '
                    f'unmodified mmdet sources from oracle/_ref), '
                    f'{ref["images"]} images 800x1344, 7 GT/img, fp32; 1 warm-up '
                    f'+ median of {ref["reps"]} steps, {st["total"]:.1f} s/step '
                    f'on {threads} threads of {model} ({phys} physical / '
                    f'{logical} logical)')
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import net_oracle as NO
    from ld_amd import build_detector, model_zoo, synthetic
    det = build_detector(model_zoo.ld_detector(sdepth, tdepth))
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2)
    torch.set_num_threads(threads)
    NO.ld_train_step(ssd, tsd, batch, sdepth, tdepth, with_backward=True)
    runs = []
    for _ in range(reps):
        tm = {}
        t0 = time.time()
        NO.ld_train_step(ssd, tsd, batch, sdepth, tdepth, with_backward=True,
                         timings=tm)
        tm['total'] = time.time() - t0
        runs.append(tm)
    runs.sort(key=lambda r: r['total'])
    med = runs[len(runs) // 2]
    return dict(value=n / med['total'], unit='images/sec', cores=threads,
                kind='port', cpu_model=model, physical_cores=phys,
                logical_cpus=logical,
                stages_s={k: round(v, 3) for k, v in med.items()},
                loss_block_s=round(med.get('loss_block', 0.0), 4),
                sample=f'FALLBACK (no oracle/_ref archive): the CPU oracle port '
                f'of the LD train step (student net + teacher net + targets/'
                f'loss block + backward), {n} images 800x1344, torch-CPU fp32 '
                f'+ numpy loss oracle; 1 warm-up + median of {reps} steps, '
                f'{med["total"]:.1f} s/step on {threads} threads of '
                f'{model} ({phys} physical / {logical} logical)')


def _device_count():
    # LD_BENCH_FAKE_DEVICES: test hook of tests/test_bench_launcher.py (the
    # launcher logic runs on a box without GPUs); never set it for a measurement
    fake = os.environ.get('LD_BENCH_FAKE_DEVICES')
    return int(fake) if fake else torch.cuda.device_count()


def launch_plan(gpus, env, n_devices, argv):
    """What `bench.py --gpus N` has to do before it may measure anything.

    Returns ('run', None) when this process is a rank of an N-rank job (or N is
    1), or ('spawn', cmd) when it was started WITHOUT a launcher and must
    re-execute itself as N ranks (the reference's tools/dist_train.sh does the
    same with torch.distributed.launch).  Raises SystemExit with a message when
    the request cannot be honoured: fewer devices than ranks, or a launcher
    whose WORLD_SIZE disagrees with --gpus -- a line that says n_gpus: N must
    have run on N GPUs."""
    if gpus < 1:
        raise SystemExit(f'bench.py: --gpus {gpus} is not a rank count')
    if n_devices < gpus:
        raise SystemExit(f'bench.py: --gpus {gpus} requested but only '
                         f'{n_devices} GPU(s) are visible on this node')
    if 'WORLD_SIZE' in env:
        world = int(env['WORLD_SIZE'])
        if world != gpus:
            raise SystemExit(f'bench.py: --gpus {gpus} but the launcher started '
                             f'WORLD_SIZE={world} ranks')
        if int(env.get('LOCAL_RANK', 0)) >= n_devices:
            raise SystemExit(f'bench.py: LOCAL_RANK {env.get("LOCAL_RANK")} has '
                             f'no device ({n_devices} visible)')
        return 'run', None
    if gpus == 1:
        return 'run', None
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return 'spawn', cmd


def pin_rank(local, local_world):
    """One Python enqueuer per GPU: give every rank its own slice of the host's
    cores (contiguous, so that it stays on one socket / NUMA node where the
    numbering allows) and a small OpenMP pool.  N ranks that each start
    os.cpu_count() OpenMP threads and migrate freely cost each other the launch
    latency the step depends on.  LD_BENCH_PIN=0 turns it off."""
    if os.environ.get('LD_BENCH_PIN', '1') != '1':
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    per = max(len(cpus) // max(local_world, 1), 1)
    mine = cpus[local * per:(local + 1) * per] or cpus
    if local_world > 1:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            return None
    threads = max(1, min(len(mine), 8))
    os.environ.setdefault('OMP_NUM_THREADS', str(threads))
    torch.set_num_threads(int(os.environ['OMP_NUM_THREADS']))
    return dict(cpus=len(mine), first=mine[0], last=mine[-1],
                omp_threads=int(os.environ['OMP_NUM_THREADS']))


def main():
    args = parse()
    action, cmd = launch_plan(args.gpus, os.environ, _device_count(), sys.argv[1:])
    if action == 'spawn':
        import subprocess
        print(f'[bench] --gpus {args.gpus} without a launcher: starting '
              f'{args.gpus} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # stdout carries exactly ONE line, the JSON record: libraries that print to
    # stdout (RCCL's version banner at communicator creation) go to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get('LD_BENCH_LAUNCH_ONLY'):  # test hook: stop after the launch checks
        print(f'[bench] launch-only rank {rank} of {world} local {local}', flush=True)
        return
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    pinned = pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f'bench.py: process group has '
                             f'{dist.get_world_size()} ranks, --gpus {args.gpus}')
    import __graft_entry__
    if not os.path.exists(os.path.join(REPO, 'ld_amd', '_lib',
                                       'libldhip.so')):
        if local == 0:
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    from ld_amd import model_zoo
    from ld_amd.train import SGDTrainer

    # The north-star kernel is timed FIRST, in the fresh process: after the
    # train legs the same launch measures 3-6 % slower (clocks / allocator state
    # after ~100 steps at full power).  It is timed
    # again after them and that figure is reported beside it.
    ldkl_first = None
    if rank == 0 and not args.no_kernel_roofline:
        try:
            ldkl_first = ldkl_roofline(dev)
        except Exception as e:  # never lose the bench line over the extra leg
            print(f'[bench] early LD-KL leg failed: {e!r}', file=sys.stderr)
        torch.cuda.empty_cache()

    build_det, workload_name, gflop_per_img = _workload(args.config)
    det = build_det(dev)
    trainer = SGDTrainer(det, lr=model_zoo.OPTIMIZER['lr'],
                         momentum=model_zoo.OPTIMIZER['momentum'],
                         weight_decay=model_zoo.OPTIMIZER['weight_decay'])
    cpu_batch, dbatch = make_batch(args.batch_per_gpu, args.num_gt,
                                   1234 + rank, dev)
    # Software pipelining of the FROZEN teacher (ld_amd/detectors.py
    # prefetch_teacher): step i trains on batch i and enqueues the teacher
    # forward of batch i + 1 on the side stream, where it runs under step i's
    # backward; step i + 1 consumes it.  Every timed step still contains exactly
    # one teacher forward (the one for the following batch) and the results are
    # bit-identical to the in-step teacher (tests/test_gpu_graph.py); what a
    # data loader that holds the next batch makes possible.  Two distinct batch
    # tensors alternate, so the queue is exercised as in a real epoch.
    prefetch = not args.no_prefetch
    _, dbatch_b = make_batch(args.batch_per_gpu, args.num_gt, 5678 + rank, dev)
    ring = [dbatch, dbatch_b]
    counter = [0]

    def one_step():
        i = counter[0]
        counter[0] = i + 1
        if prefetch:
            return trainer.step(ring[i % 2], next_data=ring[(i + 1) % 2])
        return trainer.step(ring[i % 2])

    def timed(n_warm, n_steps):
        # one priming step outside the W warm-up steps: first-call work of a
        # process (weight images, workspaces, allocator growth) must not fall
        # into the timed region when W = 0.  Conv shapes come from the shipped
        # table (ld_amd/tune/gfx950.txt), identical on every rank; nothing is
        # timed or synchronised inside the launches.
        out = one_step()
        torch.cuda.synchronize()
        for _ in range(n_warm):
            out = one_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = one_step()
        t_enq = time.perf_counter() - t0  # host time to enqueue K steps
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rank_dt[:] = [dt]
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(every, tt)
            rank_dt[:] = [float(x) for x in every]
            dt = max(rank_dt)
        return dt, t_enq, float(out['log_vars']['loss'])

    rank_dt = []
    exposed = []
    from ld_amd.train import collectives_on
    trainer.arena.exposed = exposed if collectives_on() else None
    dt, t_enq, loss_val = timed(args.warmup, args.steps)
    rank_ms = [x / args.steps * 1e3 for x in rank_dt]
    exposed_ms = None
    if exposed:
        torch.cuda.synchronize()
        ex = [a.elapsed_time(b) for a, b in exposed[-args.steps:]]
        exposed_ms = sum(ex) / len(ex)
    trainer.arena.exposed = None
    hits0 = getattr(det, 'prefetch_hits', 0)

    def synced_median(n):
        # SURVEY 8(d), the definition of `value`: every step timed on its own with
        # HIP events on the step's stream between two device synchronisations
        # (N > 1: a barrier first, so the ranks start together), the MEDIAN of
        # those, MAX over ranks.  A synchronised step cannot overlap its tail with
        # the next step's head, so this is the larger (conservative) figure; the
        # K-step bracket the driver's contract describes is reported beside it.
        ts = []
        for _ in range(n):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            one_step()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        med = ts[len(ts) // 2]
        if world > 1:
            tt = torch.tensor([med], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            med = float(tt)
        return med

    ms_sync_median = synced_median(min(max(args.steps, 3), 21))
    dt_plain = None
    if prefetch:  # the same K steps with the teacher inside each step, for the record
        prefetch = False
        dt_plain, _, _ = timed(1, args.steps)
        prefetch = True

    res = None
    parity_note = None
    if args.config == 4:
        parity_note = ('DCN with non-zero offsets: parity UNPINNED (mmcv.ops.'
                       'DeformConv2dPack is not in the image; csrc/dcn.hip is checked '
                       'against oracle/dcn_oracle.py, a restatement of mmcv 1.2.7\'s '
                       'published algorithm, and against plain conv at zero offsets)')
    if rank == 0:
        ms = dt / args.steps * 1e3
        imgs = args.batch_per_gpu * world * args.steps / dt
        res = {
            'metric': METRIC,
            # global batch / median of individually synchronised,
            # HIP-event-timed steps (MAX over ranks)
            'value': args.batch_per_gpu * world / (ms_sync_median * 1e-3),
            'unit': 'images/sec',
            'n_gpus': world,
            'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_sync_median, 'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'value_definition': (
                'global batch / median over %d steps, each timed with HIP events '
                'between two device synchronisations (SURVEY.md 8(d)); the K steps '
                'of the driver contract, enqueued back to back inside one barrier + '
                'synchronize bracket, are images_per_sec_k_step_bracket / '
                'ms_per_step_k_step_bracket below' % min(max(args.steps, 3), 21)),
            'images_per_sec_k_step_bracket': imgs,
            'ms_per_step_k_step_bracket': ms,
            'config': {
                'workload': workload_name + ', 800x1333 padded to 800x1344, '
                            f'{args.num_gt} GT/img',
                'global_batch': args.batch_per_gpu * world,
                'batch_per_gpu': args.batch_per_gpu,
                'parallelism': f'dp{world}',
                'parity_note': parity_note,
                'optimizer': 'SGD(momentum 0.9, wd 1e-4), step included',
                'last_loss': loss_val,
                'host_enqueue_ms_per_step': t_enq / args.steps * 1e3,
                'host_enqueue_note': (
                    'wall time of the host loop that enqueues the K steps; it '
                    'includes back-pressure from full hardware queues whenever the '
                    'GPU is the slower side (a C launch loop over the same ~500 '
                    'launches, step_list below, takes 7 ms on an idle queue and the '
                    'same 24 ms as Python when the fp32 step is GPU-bound: '
                    'profiles/r06_step_list_*.json)'),
                'ms_per_step_synchronised_median': ms_sync_median,
                'ms_per_step_per_rank_k_step_bracket': {
                    'min': min(rank_ms), 'max': max(rank_ms)} if rank_ms else None,
                'exposed_allreduce_ms_per_step': exposed_ms,
                'exposed_allreduce_note': (
                    'HIP-event time the compute stream spends in GradArena.finish '
                    'waiting for the bucketed gradient all-reduces that backward did '
                    'not hide (rank 0, mean over the timed steps); null without a process group'),
                'rank_cpu_placement': pinned,
                'prime_steps': 1,
                'teacher_prefetch': bool(prefetch),
                'teacher_prefetch_note': (
                    'the frozen teacher forward of batch i+1 is enqueued on '
                    'its own stream during step i and consumed by step i+1 '
                    '(bit-identical results); each timed step contains one '
                    'teacher forward; two distinct batches alternate'),
                'teacher_prefetch_hits': hits0,
                'images_per_sec_teacher_in_step': (
                    args.batch_per_gpu * world * args.steps / dt_plain
                    if dt_plain else None),
            },
        }
    # kernel-level legs: per-device figures, reported by rank 0.  The
    # instrumented steps are ordinary train steps, so with N > 1 EVERY rank runs
    # them -- their gradient / normaliser all-reduces must be matched on all
    # ranks (rank 0 alone would pair them with the other ranks' barrier).
    roof = None
    if not args.no_kernel_roofline:
        roof = kernel_roofline(trainer, dbatch, args.profile_steps)
    if rank == 0 and roof is not None:
        res['roofline'] = roof
        # step-level view of the same bound: analytic conv FLOPs per image
        # (1835.7 GFLOP) / step time
        if gflop_per_img is not None:
            res['roofline']['step_tflops_analytic'] = \
                gflop_per_img * 1e9 * args.batch_per_gpu / \
                (res['ms_per_step'] * 1e-3) / 1e12
        after = ldkl_roofline(dev)
        if ldkl_first is not None:
            ldkl_first['us_after_train_legs'] = after['us']
            ldkl_first['frac_after_train_legs'] = after['frac']
            ldkl_first['order_note'] = (
                'achieved / frac / us: timed at process start, before the '
                'train legs; *_after_train_legs: the same launches timed again '
                'after them (GPU at full power for ~100 steps)')
            res['roofline_ldkl'] = ldkl_first
        else:
            res['roofline_ldkl'] = after
    # ---- bf16 leg (BASELINE config 3's arithmetic on this rank count): the
    # same step with bf16 matrix operands.  The headline `value` above stays
    # the fp32 config; this is reported beside it.
    if not args.no_bf16:
        from ld_amd import layers as Y
        Y.set_precision('bf16')
        dtb, tenqb, lossb = timed(args.warmup, args.steps)
        ms_sync_b = synced_median(min(max(args.steps, 3), 21))
        roofb = None
        if not args.no_kernel_roofline:
            roofb = kernel_roofline(trainer, dbatch, args.profile_steps,
                                    bf16=True)
        Y.set_precision('fp32')
        if rank == 0:
            res['bf16'] = {
                'workload': 'same step, conv matrix operands in bf16 '
                            '(v_mfma_f32_32x32x16_bf16, fp32 accumulate; fp32 '
                            'master weights, activations, norms, loss block)',
                'value': args.batch_per_gpu * world / (ms_sync_b * 1e-3),
                'unit': 'images/sec', 'ms_per_step': ms_sync_b,
                'images_per_sec_k_step_bracket':
                    args.batch_per_gpu * world * args.steps / dtb,
                'ms_per_step_k_step_bracket': dtb / args.steps * 1e3,
                'host_enqueue_ms_per_step': tenqb / args.steps * 1e3,
                'last_loss': lossb, 'dtype': 'bf16 operands / f32 accumulate',
                'stepper': 'eager (Python issues every launch)',
                # round 6: the GPU side of this step fell to ~10.9 ms, the eager
                # host loop needs 9-11 ms on one core: where the two meet this leg
                # is HOST-bound (enqueue time == step time) and noisy with the host;
                # the same step re-issued by the C launch loop is step_list.bf16
                'host_bound': bool(tenqb >= 0.95 * dtb),
            }
            if roofb is not None:
                if gflop_per_img is not None:
                    roofb['step_tflops_analytic'] = \
                        gflop_per_img * 1e9 * args.batch_per_gpu / \
                        (dtb / args.steps) / 1e12
                res['roofline_bf16'] = roofb
    # ---- step-list leg: the same step captured ONCE (hipGraph capture as the
    # recorder) and re-issued by a C launch loop (ld_step_list_*, csrc/graphlist.hip),
    # teacher one step ahead inside the list, a fresh batch with a different
    # number of GT boxes per replay.  hipGraphLaunch itself is NOT used: it costs
    # ~22 us of host time per node on this runtime and ran behind the eager step in
    # every round (r05: fp32 55.3 / 56.9 vs 58.8 img/s); its legs are gone from this
    # line.  Reported beside the eager numbers; `value` stays the eager fp32 step.
    # (not with N > 1 ranks: a captured step would contain RCCL collectives, which
    # the capture refuses -- they race with ProcessGroupNCCL's watchdog)
    if not args.no_graph and world == 1 and not os.environ.get('LD_FORCE_COLLECTIVES') == '1':
        from ld_amd import layers as Y
        from ld_amd.train import PipelinedGraphedStep
        list_res = {}
        for mode in (['fp32'] if args.no_bf16 else ['fp32', 'bf16']):
            Y.set_precision(mode)
            try:
                trainer.step(dbatch)  # images of the mode exist before capture
                torch.cuda.synchronize()
                fresh = [make_batch(args.batch_per_gpu, g, 4321 + rank + g,
                                    dev)[1] for g in (5, 11, args.num_gt)]
                ps = PipelinedGraphedStep(trainer, fresh[0], fresh[1], warmup=1,
                                          launcher='list')
                for i in range(max(args.warmup, 1)):
                    ps.step(fresh[(i + 1) % len(fresh)])
                torch.cuda.synchronize()
                # host cost of ONE replay on an idle queue (no back-pressure)
                h0 = time.perf_counter()
                ps.step(fresh[0])
                host_idle = time.perf_counter() - h0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    out = ps.step(fresh[(i + 1 + args.warmup) % len(fresh)])
                t_enq_l = time.perf_counter() - t0
                torch.cuda.synchronize()
                dtp = time.perf_counter() - t0
                list_res[mode] = {
                    'value': args.batch_per_gpu * world * args.steps / dtp,
                    'unit': 'images/sec', 'ms_per_step': dtp / args.steps * 1e3,
                    'host_enqueue_ms_per_step': t_enq_l / args.steps * 1e3,
                    'host_ms_one_replay_idle_queue': host_idle * 1e3,
                    'last_loss': float(out['log_vars']['loss']),
                    'fresh_batch_per_replay': True,
                    'gt_per_image_cycle': [5, 11, args.num_gt],
                    'list': ps.lists[0].info}
                del ps
            except Exception as e:  # report, never lose the headline line
                list_res[mode] = {'error': f'{type(e).__name__}: {e}'[:300]}
        Y.set_precision('fp32')
        if rank == 0:
            res['step_list'] = list_res
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and \
            args.config == 2:  # the CPU port restates configs[1]'s nets
        res['cpu_baseline'] = cpu_baseline(cpu_batch)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
