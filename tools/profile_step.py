"""The benchmark train step alone, for rocprofv3 (--kernel-trace --stats / --pmc):
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o step -- \
        python tools/profile_step.py --mode bf16 --steps 5
Same model, batch and shipped conv shape table as bench.py; `--graph` replays the
captured hipGraph instead of enqueueing eagerly."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='fp32', choices=['fp32', 'bf16'])
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--graph', action='store_true')
    ap.add_argument('--pipeline', action='store_true',
                    help='teacher of the next batch under this step')
    ap.add_argument('--serial', action='store_true',
                    help='ONE stream: teacher inside the step on the main '
                         'stream, weight gradients on the main stream -- no two '
                         'kernels overlap, so a rocprofv3 --kernel-trace --stats '
                         'of this run sums to the serialised conv time that '
                         'bench.py\'s roofline.conv_ms_per_step reports')
    ap.add_argument('--hi-prio', action='store_true',
                    help='experiment: the whole step on a HIGH-priority stream '
                         '(the teacher / weight-gradient side streams stay at '
                         'normal priority and fill what the student chain leaves)')
    ap.add_argument('--buckets', default='',
                    help='write the gradient-bucket timeline (when each '
                         'bucket of the arena is complete, relative to '
                         'backward) here')
    ap.add_argument('--layers', default='',
                    help='write the per-layer conv table (HIP events around '
                         'every conv launch, teacher on the main stream) here')
    args = ap.parse_args()
    from ld_amd import layers as Y
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import GraphedStep, SGDTrainer
    dev = torch.device('cuda:0')
    if 'RANK' in os.environ:  # under torch.distributed.run: RCCL group
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(dev)
        dist.init_process_group('nccl', device_id=dev)
    Y.set_precision(args.mode)
    if args.hi_prio:
        hi = torch.cuda.Stream(dev, priority=-1)
        hi.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(hi)
    if args.serial:
        Y._WGRAD_STREAM[0] = False
    det = model_zoo.build_seeded_ld_detector(50, 101, dev)
    if args.serial:
        det.use_teacher_stream = False
    tr = SGDTrainer(det, lr=model_zoo.OPTIMIZER['lr'])
    b = synthetic.synthetic_batch(2, (800, 1333), (800, 1344), 7, 1234)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    for _ in range(args.warmup):
        tr.step(d)
    torch.cuda.synchronize()
    if args.graph:
        g = GraphedStep(tr, d)
        run = g.replay
    else:
        if args.pipeline:
            # three DISTINCT batch tensors in rotation: step i trains on batch
            # i and enqueues the teacher of batch i + 1 under it
            ring = [d]
            for sd in (77, 78):
                bb = synthetic.synthetic_batch(2, (800, 1333), (800, 1344), 7, sd)
                ring.append(dict(img=bb['img'].to(dev), img_metas=bb['img_metas'],
                                 gt_bboxes=[x.to(dev) for x in bb['gt_bboxes']],
                                 gt_labels=[x.to(dev) for x in bb['gt_labels']]))
            state = dict(i=0)

            def run():
                i = state['i']
                state['i'] = i + 1
                return tr.step(ring[i % 3], next_data=ring[(i + 1) % 3])
            for _ in range(3):
                run()
            torch.cuda.synchronize()
        else:
            run = lambda: tr.step(d)  # noqa: E731
    import time
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f'{args.mode} graph={args.graph} pipeline={args.pipeline}: '
          f'{dt * 1e3:.2f} ms/step, '
          f'{2 / dt:.1f} img/s'
          + (f' (teacher prefetch hits {getattr(det, "prefetch_hits", 0)})'
             if args.pipeline else ''))
    if args.mode == 'bf16':
        Y.C8_STATS.update(converted=0, reused=0)
        run()
        print('C8 operand images per step:', Y.C8_STATS)
    if args.buckets:
        # How much of backward is left to hide each bucket's all-reduce behind.
        # (On ONE rank RCCL launches no kernel for an in-place all-reduce --
        # profiles/r03_overlap_trace_1rank_no_rccl_kernels.txt -- so kernel
        # concurrency itself can only be traced on a multi-GPU node.)
        rows = []
        for _ in range(args.steps):
            tr.arena.trace = []
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            tr.arena.zero_grad()
            head = det.bbox_head
            head.unit_upstream = True
            losses = det(**d)
            loss, _ = det._parse_losses(losses)
            e0.record()
            loss.backward()
            e1.record()
            head.unit_upstream = False
            tr.arena.finish()
            Y.sgd_step(tr.arena.flat_param, tr.arena.flat_grad,
                       tr.flat_momentum, tr.lr, tr.momentum, tr.weight_decay)
            e2.record()
            torch.cuda.synchronize()
            bw = e0.elapsed_time(e1)
            rows.append((bw, [(b, e0.elapsed_time(ev))
                              for b, ev in tr.arena.trace]))
        tr.arena.trace = None
        with open(args.buckets, 'w') as f:
            f.write('# gradient-bucket timeline, %s, C2 step: bucket, MiB, '
                    'ready at (ms after backward starts), backward left (ms), '
                    'fraction of backward left\n' % args.mode)
            bw = sorted(r[0] for r in rows)[len(rows) // 2]
            f.write('# backward = %.2f ms (median of %d steps); ring all-reduce '
                    'of B bytes over 8 GPUs moves 2*7/8*B per GPU: at ~300 '
                    'GB/s effective a 32 MiB bucket needs ~0.2 ms\n'
                    % (bw, len(rows)))
            last = rows[-1][1]
            for b, t in last:
                bk = tr.arena.buckets[b]
                mib = (bk['end'] - bk['start']) * 4 / 2**20
                f.write('%d,%.1f,%.2f,%.2f,%.3f\n'
                        % (b, mib, t, rows[-1][0] - t, 1 - t / rows[-1][0]))
        print(open(args.buckets).read())
    if args.layers:
        det.use_teacher_stream = False
        tr.step(d)
        with Y.KernelProfile() as prof:
            for _ in range(args.steps):
                tr.step(d)
        rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1][0])
        tot = sum(v[0] for _, v in rows)
        with open(args.layers, 'w') as f:
            f.write(f'# per-layer conv table, {args.mode}, {args.steps} steps; '
                    'ms = per step over all launches of that shape\n')
            f.write('kind,shape,launches_per_step,ms_per_step,tflops,'
                    'pct_of_conv\n')
            for (tag, key), (t, fl, n) in rows:
                f.write(f'{tag},{key},{n / args.steps:g},'
                        f'{t / args.steps * 1e3:.4f},{fl / t / 1e12:.1f},'
                        f'{100 * t / tot:.2f}\n')
        print('conv total %.2f ms/step; wrote %s' %
              (tot / args.steps * 1e3, args.layers))
    if 'RANK' in os.environ:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
