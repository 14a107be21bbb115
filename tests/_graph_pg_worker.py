"""Worker of tests/test_gpu_graph_pg.py (own process: the runtime's queue
configuration is read at the first HIP call).  An RCCL process group of ONE rank
with every collective forced (LD_FORCE_COLLECTIVES=1), 8 hardware queues and the
graph executor held to 2 streams -- what importing ld_amd sets up in a
multi-process job.  Runs a batch sequence (different GT counts, two padded
shapes) through eager ``SGDTrainer.step`` and through ``AutoStepper`` in its
DEFAULT mode for the precision (bf16: one captured hipGraph per shape, bucket
all-reduces inside the capture) and prints one JSON line."""
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
        for d in seq:
            out = stepper(d)
            losses.append(float(out['log_vars']['loss']))
        torch.cuda.synchronize()
        return tr, losses

    tr0, l0 = run(lambda tr: tr.step)
    eager_calls = calls['all_reduce']
    calls['all_reduce'] = 0
    auto = {}

    def make(tr):
        auto['s'] = T.AutoStepper(tr)
        return auto['s'].step

    tr1, l1 = run(make)
    res = dict(
        precision=precision, mode=auto['s'].mode, captures=auto['s'].captures,
        collectives_on=bool(T.collectives_on()), graph_queues_ok=bool(T.graph_queues_ok()),
        hwq=os.environ.get('GPU_MAX_HW_QUEUES'),
        graph_queues=os.environ.get('DEBUG_HIP_FORCE_GRAPH_QUEUES'),
        buckets=len(tr1.arena.buckets), eager_all_reduce_calls=eager_calls,
        auto_all_reduce_calls=calls['all_reduce'],
        params_equal=bool(torch.equal(tr0.arena.flat_param, tr1.arena.flat_param)),
        momentum_equal=bool(torch.equal(tr0.flat_momentum, tr1.flat_momentum)),
        losses_equal=l0 == l1, losses=l1)
    # replay vs eager time of this small step, same process (informational)
    s = auto['s']
    d = seq[0]
    for _ in range(3):
        s.step(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        s.step(d)
    torch.cuda.synchronize()
    res['auto_ms_per_step'] = (time.perf_counter() - t0) * 100
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
