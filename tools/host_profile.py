"""cProfile of the host side of eager train steps (where the Python time of the
~500 launches per step goes).  The autograd engine is held to the calling thread
(torch.autograd.set_multithreading_enabled(False)) so that the backward's Python
shows up in the profile.
    python tools/host_profile.py [fp32|bf16] [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ld_amd import layers as Y  # noqa: E402
from ld_amd import model_zoo  # noqa: E402
from ld_amd.train import SGDTrainer  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device('cuda:0')
    Y.set_precision(mode)
    det = model_zoo.build_seeded_ld_detector(50, 101, dev)
    tr = SGDTrainer(det, lr=model_zoo.OPTIMIZER['lr'])
    b = [bench.make_batch(2, g, 4321 + g, dev)[1] for g in (7, 5)]
    for i in range(6):
        tr.step(b[i % 2], next_data=b[(i + 1) % 2])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):
        for i in range(2):
            tr.step(b[i % 2], next_data=b[(i + 1) % 2])
        torch.cuda.synchronize()
        pr.enable()
        for i in range(steps):
            tr.step(b[i % 2], next_data=b[(i + 1) % 2])
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime')
    print(f'# {steps} eager {mode} steps; times are totals over the steps')
    st.print_stats(60)
    st.sort_stats('cumulative')
    st.print_stats(35)


if __name__ == '__main__':
    main()
