"""Do concurrent streams fill the ramp / store phases of the small conv launches?
The same 16 launches of one layer captured into a hipGraph (no host enqueue
time in the measurement) on 1, 2 and 4 streams; aggregate TFLOP/s of each.
DESIGN.md section 7 items 1 / 2 lean on this: a 1x1 layer alone reaches ~85-95
TFLOP/s, the overlapped train step runs its convs at ~100 on average.

    python tools/probe/run_concurrency.py          (through gpurun)
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

# name, N, Cin, Cout, k, pad, (H, W), residual + affine + relu epilogue
LAYERS = [('1x1 256>1024 50x84 +bn+res+relu', 2, 256, 1024, 1, 0, (50, 84), True),
          ('1x1 1024>256 50x84 +bn+relu', 2, 1024, 256, 1, 0, (50, 84), False),
          ('3x3 256>256 50x84 +bn+relu', 2, 256, 256, 3, 1, (50, 84), False),
          ('1x1 128>512 100x168 +bn+res+relu', 2, 128, 512, 1, 0, (100, 168), True)]
LAUNCHES = 16


def main():
    from ld_amd import layers as Y
    dev = torch.device('cuda:0')
    out = []
    for name, N, cin, cout, k, pad, hw, res in LAYERS:
        P = hw[0] * hw[1]
        g = torch.Generator().manual_seed(cin + cout + k)
        x = torch.randn(N, cin, P, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k)**0.5).to(dev)
        scale = (torch.rand(cout, generator=g) + .5).to(dev)
        shift = torch.randn(cout, generator=g).to(dev)
        r = torch.randn(N, cout, P, generator=g).to(dev) if res else None
        flops = 2.0 * N * P * cin * cout * k * k

        def one():
            return Y.conv_forward_raw(x, w, 1, pad, (hw, ), scale=scale, shift=shift,
                                      residual=r, relu=True)[0]
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        row = dict(layer=name, gflop=flops / 1e9)
        for nstreams in (1, 2, 4):
            streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
            cap = torch.cuda.Stream(device=dev)
            cap.wait_stream(torch.cuda.current_stream(dev))
            graph = torch.cuda.CUDAGraph()
            keep = []
            with torch.cuda.graph(graph, stream=cap):
                for s in streams:
                    s.wait_stream(cap)
                for i in range(LAUNCHES):
                    with torch.cuda.stream(streams[i % nstreams]):
                        keep.append(one())
                for s in streams:
                    cap.wait_stream(s)
            torch.cuda.synchronize()
            for _ in range(2):
                graph.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(True), torch.cuda.Event(True)
            a.record()
            for _ in range(5):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            dt = a.elapsed_time(b) * 1e-3 / 5 / LAUNCHES
            row[f'streams_{nstreams}'] = dict(us_per_launch=dt * 1e6,
                                              tflops=flops / dt / 1e12)
            del graph, keep
        print(name, {k2: round(v['tflops'], 1) for k2, v in row.items()
                     if isinstance(v, dict)}, flush=True)
        out.append(row)
    path = os.path.join(REPO, 'gpurun_out', 'probe_concurrency.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
