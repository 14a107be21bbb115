"""The captured train step re-issued as a launch list (ld_step_list_*,
csrc/graphlist.hip) against eager steps and hipGraph replays: ms / step, host
enqueue time, at the bench configuration.
    python tools/bench_step_list.py [fp32|bf16] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo  # noqa: E402
from ld_amd.train import PipelinedGraphedStep, GraphedStep, SGDTrainer  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device('cuda:0')
    Y.set_precision(mode)
    det = model_zoo.build_seeded_ld_detector(50, 101, dev)
    tr = SGDTrainer(det, lr=model_zoo.OPTIMIZER['lr'])
    fresh = [bench.make_batch(2, g, 4321 + g, dev)[1] for g in (7, 5, 11)]
    res = {}

    def timeit(name, fn, warm=5):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = fn(i + warm)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = dict(ms_per_step=dt / steps * 1e3, host_enqueue_ms=t_enq / steps * 1e3,
                         img_per_s=2 * steps / dt, loss=float(out['log_vars']['loss']))
        print(name, res[name], flush=True)

    timeit('eager_prefetch', lambda i: tr.step(fresh[i % 3], next_data=fresh[(i + 1) % 3]))
    for launcher in ('list', 'graph'):
        try:
            ps = PipelinedGraphedStep(tr, fresh[0], fresh[1], warmup=1, launcher=launcher)
            if launcher == 'list':
                print('list info', [l.info for l in ps.lists], flush=True)
                res['list_info'] = ps.lists[0].info
            timeit(f'pipelined_{launcher}', lambda i: ps.step(fresh[(i + 1) % 3]))
            del ps
        except Exception as e:
            print(launcher, 'failed:', repr(e)[:400], flush=True)
    try:
        gs = GraphedStep(tr, fresh[0], warmup=1, launcher='list')
        print('single list info', gs.list.info, flush=True)

        def one(i):
            gs.copy_inputs(fresh[i % 3])
            return gs.replay()
        timeit('single_list', one)
    except Exception as e:
        print('single list failed:', repr(e)[:400], flush=True)
    if len(sys.argv) > 3:
        json.dump(dict(mode=mode, steps=steps, results=res), open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
