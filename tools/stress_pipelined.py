"""Stress of tests/test_gpu_graph.py::test_pipelined_graphed_step_equals_eager[bf16]
(round 5 saw it fail intermittently with NaN in the tail of the parameter arena):
the scenario K times in one process; on a mismatch prints which parameters differ.
    python tools/stress_pipelined.py [K] [bf16|fp32]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from ld_amd import layers as Y  # noqa: E402
from ld_amd.train import PipelinedGraphedStep  # noqa: E402
import test_gpu_graph as TG  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dev = torch.device('cuda:0')
Y.set_precision(mode)
bad = 0
for it in range(K):
    b = [TG._batch_g(31 + i, g, dev) for i, g in enumerate(([3, 2], [5, 1], [2, 7], [4, 4], [1, 1]))]
    eager = TG._trainer(dev)
    for d in [b[0], b[1], b[0], b[1], b[2], b[3]]:
        out_e = eager.step(d)
    torch.cuda.synchronize()
    tr = TG._trainer(dev)
    ps = PipelinedGraphedStep(tr, b[0], b[1], warmup=1, max_gt=16)
    outs = [ps.step(b[1]), ps.step(b[2]), ps.step(b[3]), ps.step(b[4])]
    torch.cuda.synchronize()
    ok = torch.equal(tr.arena.flat_param, eager.arena.flat_param)
    if not ok:
        bad += 1
        names = {id(p): k for k, p in tr.model.named_parameters()}
        diff = []
        for p, o in zip(tr.arena.order, tr.arena.offsets):
            a = tr.arena.flat_param[o:o + p.numel()]
            e = eager.arena.flat_param[o:o + p.numel()]
            if not torch.equal(a, e):
                diff.append((names[id(p)], tr.arena.bucket_of[id(p)],
                             bool(torch.isnan(a).any()), float((a - e).abs().nan_to_num(1e9).max())))
        print(f'iter {it}: {len(diff)} of {len(tr.arena.order)} parameters differ; buckets '
              f'{sorted({d[1] for d in diff})} of {len(tr.arena.buckets)}; first 6: {diff[:6]}; '
              f'last 3: {diff[-3:]}', flush=True)
    else:
        print(f'iter {it}: ok', flush=True)
    del ps, tr, eager
print(f'{bad} of {K} iterations mismatched ({mode}); env '
      f'{ {k: os.environ.get(k) for k in ("LD_DEFER_GRADS", "LD_TEACHER_REPLAY", "LD_FAN_FUSE", "LD_FAN_INPLACE", "LD_FUSED_BOTTLENECK")} }')
