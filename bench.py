#!/usr/bin/env python
"""bench.py -- images/sec of the LD train step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

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
  cpu_baseline  the CPU oracle ("port": torch-CPU nets + numpy loss block) on
                the host cores, one step of the same batch.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = 'images/sec (1333x800) GFocal-R50<-R101 LD train step'
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW"


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
    return ap.parse_args()


def make_batch(bs, num_gt, seed, dev):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(bs, (800, 1333), (800, 1344), num_gt, seed)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    return b, d


def kernel_roofline(trainer, dbatch, steps):
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
    model.use_teacher_stream = prev
    tot_t = sum(v[0] for v in agg.values())
    tot_f = sum(v[1] for v in agg.values())
    tot_n = sum(v[2] for v in agg.values())
    ach = tot_f / tot_t / 1e12 if tot_t > 0 else 0.0
    return dict(
        kernel='conv (fp32 MFMA 32x32x2 implicit GEMM: streaming fwd/dgrad + wave-private wgrad)',
        bound='mfma', achieved=ach, peak=PEAK_FP32_MFMA_TFLOPS,
        unit='TFLOP/s', frac=ach / PEAK_FP32_MFMA_TFLOPS,
        # PMC passes on the largest launch of the step (head-tower forward,
        # 52.8 GFLOP, 48.3 MB in + 45.9 MB out algorithmic): raw FETCH_SIZE
        # 196.5 MB (L2 fabric side, Infinity-Cache hits included: each of the 8
        # XCD L2s pulls the image) + WRITE_SIZE 45.9 MB
        traffic=242.4e6,
        traffic_source='profiles/r01_pmc_stream (separate --pmc passes, '
                       'head-tower forward launch, bytes per launch)',
        launches_per_step=tot_n / steps,
        avg_launch_us=tot_t / max(tot_n, 1) * 1e6,
        conv_ms_per_step=tot_t / steps * 1e3,
        gflop_per_step=tot_f / steps / 1e9,
        by_kind={k: dict(ms_per_step=v[0] / steps * 1e3,
                         tflops=v[1] / v[0] / 1e12 if v[0] else 0.0,
                         launches=v[2] / steps) for k, v in agg.items()})


def ldkl_roofline(dev):
    from ld_amd import lossblock as LB
    rows = 1 << 22  # anchors -> 2^24 anchor-side rows
    s = torch.randn(68, rows, device=dev) * 3
    t = torch.randn(68, rows, device=dev) * 3
    w = torch.rand(rows, device=dev)
    for _ in range(3):
        LB.kl_integral_dense(s, t, w, 10.0, 1.0, True)
    torch.cuda.synchronize()
    iters = 11
    evs = [(torch.cuda.Event(enable_timing=True),
            torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        LB.kl_integral_dense(s, t, w, 10.0, 1.0, True)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
    dt = ts[len(ts) // 2]  # median launch duration
    # algorithmic bytes per anchor-side row: 136 logits in + 4 weight + 4
    # integral out + 4 loss out + 68 grad out = 216 B
    nbytes = rows * 4 * 216.0
    ach = nbytes / dt / 1e9
    return dict(kernel='kl_integral_dense (fused LD-KL + Integral fwd+grad)',
                bound='hbm', achieved=ach, peak=PEAK_HBM_GBPS, unit='GB/s',
                frac=ach / PEAK_HBM_GBPS,
                # PMC pass (profiles/r01_pmc_traffic/kl_*.csv): WRITE_SIZE
                # 1 245 184 KB + 2 x FETCH_SIZE 1 122 379 KB (gfx950 FETCH_SIZE
                # counts half the bytes of a coalesced stream,
                # MI355X_MICROARCH.md "HBM") = 3.57 GB per launch vs 3.62 GB
                # algorithmic: no wasted re-reads
                traffic=(1245184 + 2 * 1122379) * 1024.0,
                traffic_source='profiles/r01_pmc_traffic (separate --pmc '
                               'passes, same kernel and size)',
                rows=rows * 4, bytes_per_row=216, us=dt * 1e6)


def cpu_baseline(batch, sdepth=50, tdepth=101):
    """The oracle ("port") timed on the host cores: one full LD step."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import net_oracle as NO
    from ld_amd import build_detector, model_zoo, synthetic
    det = build_detector(model_zoo.ld_detector(sdepth, tdepth))
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2)
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    small = dict(batch)
    t0 = time.time()
    NO.ld_train_step(ssd, tsd, small, sdepth, tdepth, with_backward=True)
    dt = time.time() - t0
    n = batch['img'].shape[0]
    return dict(value=n / dt, unit='images/sec', cores=threads, kind='port',
                sample=f'1 LD train step (fwd+loss+bwd), {n} images '
                f'800x1344, torch-CPU fp32 + numpy loss oracle, no warm-up, '
                f'{dt:.1f} s on {threads} of {cores} host threads')


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=dev)
    import __graft_entry__
    if not os.path.exists(os.path.join(REPO, 'ld_amd', '_lib',
                                       'libldhip.so')):
        if local == 0:
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    from ld_amd import model_zoo
    from ld_amd.train import SGDTrainer

    det = model_zoo.build_seeded_ld_detector(50, 101, dev)
    trainer = SGDTrainer(det, lr=model_zoo.OPTIMIZER['lr'],
                         momentum=model_zoo.OPTIMIZER['momentum'],
                         weight_decay=model_zoo.OPTIMIZER['weight_decay'])
    cpu_batch, dbatch = make_batch(args.batch_per_gpu, args.num_gt,
                                   1234 + rank, dev)

    # one priming step outside the W warm-up steps: the conv library times its
    # register-tile candidates on the first launch of every layer geometry
    # (cudnn.benchmark-style, ld_amd/csrc/conv.hip) -- a one-off per process,
    # like compilation, that must not fall into the timed region when W = 0
    out = trainer.step(dbatch)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = trainer.step(dbatch)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(dbatch)
    t_enq = time.perf_counter() - t0  # host time to enqueue K steps
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss_val = float(out['log_vars']['loss'])

    res = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        imgs = args.batch_per_gpu * world * args.steps / dt
        res = {
            'metric': METRIC, 'value': imgs, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'ld_r50_gflv1_r101_fpn_coco_1x (BASELINE.json '
                            'configs[1]): GFocal-R50 student <- R101 teacher, '
                            'fp32, 800x1333 padded to 800x1344, '
                            f'{args.num_gt} GT/img',
                'global_batch': args.batch_per_gpu * world,
                'batch_per_gpu': args.batch_per_gpu,
                'parallelism': f'dp{world}',
                'optimizer': 'SGD(momentum 0.9, wd 1e-4), step included',
                'last_loss': loss_val,
                'host_enqueue_ms_per_step': t_enq / args.steps * 1e3,
                'prime_steps': 1,
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
        res['roofline']['step_tflops_analytic'] = \
            1835.7e9 * args.batch_per_gpu / (res['ms_per_step'] * 1e-3) / 1e12
        res['roofline_ldkl'] = ldkl_roofline(dev)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(cpu_batch)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
