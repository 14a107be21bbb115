"""The fused frozen-teacher bottleneck (csrc/conv_fused.hip) against the three
launches it replaces, on the R101 layer3 shape (2 x 1024 x 50 x 84, bf16 C8), timed
in ONE process the way tools/probe/run_fp32_lds_tile.py did for fp32: HIP events
around 50 back-to-back blocks, 22 distinct blocks in rotation (the teacher's
layer3: weights do not stay in L2 between launches of the same block).
    python tools/bench_fused_block.py [out.json]     (through gpurun)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from ld_amd import layers as Y  # noqa: E402
from test_gpu_fused_block import _block  # noqa: E402

dev = torch.device('cuda:0')
Y.set_precision('bf16')
N, H, W = 2, 50, 84
blocks = [_block(dev, 100 + i) for i in range(22)]
x = torch.randn(N, 1024, H * W, device=dev).relu()
res = {}
with torch.no_grad(), Y.c8_only_scope():
    x8 = Y.C8Act(Y.to_c8(x), x.shape)
    for name, flag in (('three_launches', False), ('fused', True), ('three_launches_again', False),
                       ('fused_again', True)):
        Y._FUSED_BLOCK[0] = flag

        def chain():
            y = x8
            for b in blocks:
                y, _ = b.forward3(y, ((H, W), ))
            return y
        for _ in range(3):
            out = chain()
        torch.cuda.synchronize()
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(3):
                out = chain()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (3 * len(blocks))
            best = us if best is None else min(best, us)
        flop = 2.0 * N * H * W * (1024 * 256 * 2 + 256 * 256 * 9)
        res[name] = dict(us_per_block=best, tflops=flop / best / 1e6,
                         checksum=float(out.float().double().sum()))
        print(name, res[name], flush=True)
Y._FUSED_BLOCK[0] = True
res['bit_identical_chain'] = res['fused']['checksum'] == res['three_launches']['checksum']
res['shape'] = [N, 1024, H, W]
res['blocks_in_rotation'] = len(blocks)
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r05_fused_block.json'
json.dump(res, open(path, 'w'), indent=1)
print(json.dumps(res))
