"""The 8-wave LDS-DMA C8 kernel (csrc/conv_t256.hip) against the shipped 4-wave
C8 tile kernel on one geometry: bit-identity of the outputs and HIP-event timing.
    python tools/bench_t256.py [head|fpn|l2|l3] [out.json]
Shapes are LD_CONV_C8_SHAPE strings (BM/32 x BN/32 x NST x BK x SCH; NST = 8 =
the LDS-DMA kernel)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402

GEO = {
    # name: (N, cin, cout, k, stride, pad, levels)
    'head': (2, 256, 256, 3, 1, 1, ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))),
    'fpn': (2, 256, 256, 3, 1, 1, ((100, 168), )),
    'l2': (2, 128, 128, 3, 1, 1, ((100, 168), )),
    'l3': (2, 256, 256, 3, 1, 1, ((50, 84), )),
    'l3e': (2, 256, 1024, 1, 1, 0, ((50, 84), )),
    'l3r': (2, 1024, 256, 1, 1, 0, ((50, 84), )),
}
SHAPES = {
    'head': ['4x4x4x32x1', '4x4x2x64', '8x8x8x64', '8x6x8x64', '4x8x8x64'],
    'fpn': ['4x4x4x32x1', '4x4x2x64', '8x8x8x64', '8x6x8x64', '4x8x8x64'],
    'l2': ['2x2x2x64', '4x4x2x64', '4x8x8x64'],
    'l3': ['2x2x2x64', '4x4x2x64', '8x8x8x64', '8x6x8x64', '4x8x8x64'],
    'l3e': ['2x4x2', '4x4x2x64', '8x8x8x64', '8x6x8x64', '4x8x8x64'],
    'l3r': ['2x2x2x64', '4x4x2x64', '8x8x8x64', '8x6x8x64', '4x8x8x64'],
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'head'
    N, cin, cout, k, s, p, levels = GEO[which]
    dev = torch.device('cuda:0')
    Y.set_precision('bf16')
    Y.set_c8(True)
    P = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, cin, P, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5).to(dev)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(cout, generator=g).to(dev)
    x8 = Y.C8Act(Y.to_c8(x), x.shape)
    flops = 2.0 * N * P * cout * cin * k * k
    res, ref = [], None
    for shape in SHAPES[which]:
        os.environ['LD_CONV_C8_SHAPE'] = shape

        def run(emit):
            return Y.conv_forward_raw(x8, w, s, p, levels, scale=scale, shift=shift,
                                      relu=True, emit_c8=emit)[0]
        y = run(True)
        torch.cuda.synchronize()
        img = Y._c8_cached(y)
        if ref is None:
            ref = (y.clone(), img.clone())
            same = True
        else:
            same = bool(torch.equal(y, ref[0]) and torch.equal(img, ref[1]))
        row = dict(shape=shape, bit_identical_to_first=same)
        for emit in (False, True):
            for _ in range(3):
                run(emit)
            torch.cuda.synchronize()
            # ten launches per replay of a captured graph: no host time between them
            graph = torch.cuda.CUDAGraph()
            keep = []
            with torch.cuda.graph(graph):
                for _ in range(10):
                    keep.append(run(emit))
            graph.replay()
            torch.cuda.synchronize()
            best = None
            for _ in range(5):
                a, b = torch.cuda.Event(True), torch.cuda.Event(True)
                a.record()
                graph.replay()
                b.record()
                torch.cuda.synchronize()
                dt = a.elapsed_time(b) * 1e-3 / 10
                best = dt if best is None or dt < best else best
            del graph, keep
            row['us_c8out' if emit else 'us'] = best * 1e6
            row['tflops_c8out' if emit else 'tflops'] = flops / best / 1e12
        res.append(row)
        print(row, flush=True)
    if len(sys.argv) > 2:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
        json.dump(dict(geometry=which, N=N, cin=cin, cout=cout, k=k, levels=levels,
                       gflop=flops / 1e9, results=res), open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
