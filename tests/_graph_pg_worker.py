"""Worker of tests/test_gpu_graph_pg.py (own process: the runtime's queue
configuration is read at the first HIP call).  An RCCL process group of ONE rank
with every collective forced (LD_FORCE_COLLECTIVES=1): what importing ld_amd sets
up in a multi-process job, which stepper AutoStepper picks there, that its steps
equal plain SGDTrainer.step bit for bit over a batch sequence with two padded shapes
and changing GT counts, that capturing collectives is refused, and -- without
collectives in the capture -- that a hipGraph replay keeps its speed with the 8
hardware queues a process group needs.  Prints one JSON line."""
import json
import os
import sys
import time

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ['RANK'] = '0'
os.environ['WORLD_SIZE'] = '1'
os.environ['LD_FORCE_COLLECTIVES'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import ld_amd  # noqa: E402,F401  (sets GPU_MAX_HW_QUEUES / DEBUG_HIP_FORCE_GRAPH_QUEUES)
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo, synthetic  # noqa: E402
from ld_amd import train as T  # noqa: E402


def batch(seed, shape, gts, dev):
    b = synthetic.synthetic_batch(2, shape, shape, gts, seed)
    return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                gt_labels=[x.to(dev) for x in b['gt_labels']])


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    Y.set_precision(precision)
    seq = [batch(11, (128, 160), [3, 2], dev), batch(12, (128, 160), [1, 5], dev),
           batch(13, (160, 128), [2, 2], dev), batch(14, (128, 160), [4, 1], dev),
           batch(15, (160, 128), [6, 3], dev), batch(16, (128, 160), [2, 7], dev)]
    calls = dict(all_reduce=0)
    real = dist.all_reduce

    def counted(*a, **k):
        calls['all_reduce'] += 1
        return real(*a, **k)

    dist.all_reduce = counted

    def run(make_stepper):
        det = model_zoo.build_seeded_ld_detector(18, 18, dev, loss_im_weight=2.0)
        tr = T.SGDTrainer(det, lr=0.01, bucket_bytes=4 << 20)
        stepper = make_stepper(tr)
        losses = []
        for i, d in enumerate(seq):
            out = stepper(d, seq[i + 1] if i + 1 < len(seq) else None)
            losses.append(float(out['log_vars']['loss']))
        torch.cuda.synchronize()
        return tr, losses

    tr0, l0 = run(lambda tr: (lambda d, nxt: tr.step(d)))
    eager_calls = calls['all_reduce']
    calls['all_reduce'] = 0
    auto = {}

    def make(tr):
        auto['s'] = T.AutoStepper(tr)
        return lambda d, nxt: auto['s'].step(d, next_data=nxt)

    tr1, l1 = run(make)
    res = dict(
        precision=precision, mode=auto['s'].mode,
        collectives_on=bool(T.collectives_on()), graph_queues_ok=bool(T.graph_queues_ok()),
        hwq=os.environ.get('GPU_MAX_HW_QUEUES'),
        graph_queues=os.environ.get('DEBUG_HIP_FORCE_GRAPH_QUEUES'),
        buckets=len(tr1.arena.buckets), eager_all_reduce_calls=eager_calls,
        auto_all_reduce_calls=calls['all_reduce'],
        params_equal=bool(torch.equal(tr0.arena.flat_param, tr1.arena.flat_param)),
        momentum_equal=bool(torch.equal(tr0.flat_momentum, tr1.flat_momentum)),
        losses_equal=l0 == l1, teacher_prefetch_hits=getattr(tr1.model, 'prefetch_hits', 0))
    try:
        T.GraphedStep(tr1, seq[0])
        res['capture_refused'] = False
    except RuntimeError as e:
        res['capture_refused'] = 'refused' in str(e)
    # a capture WITHOUT collectives (suspended): replay speed under 8 hardware queues
    d = seq[0]
    with T.suspend_collectives():
        g = T.GraphedStep(tr1, d, warmup=2, warmup_collectives=False)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        res['graph_ms_per_step'] = (time.perf_counter() - t0) * 100
        for _ in range(3):
            tr1.step(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr1.step(d)
        torch.cuda.synchronize()
        res['eager_ms_per_step'] = (time.perf_counter() - t0) * 100
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
