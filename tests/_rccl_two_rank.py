"""Worker of tests/test_gpu_rccl.py::test_two_ranks_on_one_gpu_match_single_process:
two torch.distributed.run ranks that share cuda:0, each stepping on its own
half of a 4-image batch over RCCL.  Rank r saves its parameter arena to
$LD_RCCL_OUT/rank{r}.npy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    rank = int(os.environ['RANK'])
    world = int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    from test_gpu_rccl import _batch, _one_step
    data = _batch(4, 33, dev, lo=2 * rank, hi=2 * rank + 2)
    tr, _ = _one_step(dev, data, steps=1)
    np.save(os.path.join(os.environ['LD_RCCL_OUT'], f'rank{rank}.npy'),
            tr.arena.flat_param.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
