"""hipGraph replay time of the train step under a given runtime configuration
(why does a replay collapse with GPU_MAX_HW_QUEUES > 4?).
The runtime knobs are read from the ENVIRONMENT before the first HIP call:
GPU_MAX_HW_QUEUES, DEBUG_HIP_FORCE_GRAPH_QUEUES (how many internal streams a
graph launch spreads its branches over), LD_WGRAD_STREAM (0 = weight gradients
stay on the main stream inside the capture, i.e. fewer fork / join edges).
    python tools/graph_queues.py bf16|fp32 [--dot out.dot] [--pipelined]
prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo  # noqa: E402
from ld_amd.train import GraphedStep, PipelinedGraphedStep, SGDTrainer  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dot = sys.argv[sys.argv.index('--dot') + 1] if '--dot' in sys.argv else None
pipelined = '--pipelined' in sys.argv
steps = 20
dev = torch.device('cuda:0')
Y.set_precision(mode)
det = model_zoo.build_seeded_ld_detector(50, 101, dev)
tr = SGDTrainer(det, lr=0.0025)
_, d0 = bench.make_batch(2, 7, 1234, dev)
_, d1 = bench.make_batch(2, 5, 99, dev)
tr.step(d0)
torch.cuda.synchronize()
res = dict(mode=mode, pipelined=pipelined,
           env={k: os.environ.get(k) for k in (
               'GPU_MAX_HW_QUEUES', 'DEBUG_HIP_FORCE_GRAPH_QUEUES', 'LD_WGRAD_STREAM',
               'DEBUG_HIP_GRAPH_BATCH_SIZE', 'DEBUG_CLR_GRAPH_PACKET_CAPTURE',
               'LD_SHARE_SIDE_STREAMS')})
if pipelined:
    g = PipelinedGraphedStep(tr, d0, d1, warmup=1)
    run = lambda i: g.step(d1 if i % 2 == 0 else d0)  # noqa: E731
else:
    g = GraphedStep(tr, d0, warmup=2)
    if dot:
        # torch can only dump a graph captured in debug mode: capture a second one
        g2 = torch.cuda.CUDAGraph()
        g2.enable_debug_mode()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.graph(g2, stream=side):
            tr.step(g.data)
        g2.debug_dump(dot)
        txt = open(dot).read()
        res['dot_nodes'] = txt.count('[')
        res['dot_edges'] = txt.count('->')
        del g2
    run = lambda i: (g.copy_inputs(d0 if i % 2 else d1), g.replay())  # noqa: E731
for i in range(4):
    run(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    run(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
res['replay_ms_per_step'] = (t2 - t0) / steps * 1e3
res['host_ms_per_step'] = (t1 - t0) / steps * 1e3
# the eager step in the same process, teacher in the step (what one graph holds)
for _ in range(3):
    tr.step(d0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(d0 if i % 2 else d1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
res['eager_ms_per_step'] = (t2 - t0) / steps * 1e3
res['eager_host_ms_per_step'] = (t1 - t0) / steps * 1e3
print(json.dumps(res), flush=True)
