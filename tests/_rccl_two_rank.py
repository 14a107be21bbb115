"""Worker of tests/test_gpu_rccl.py::test_two_ranks_on_one_gpu_match_single_process
(two torch.distributed.run ranks that share cuda:0) and ::test_multi_gpu_ranks_
match_single_process (LD_RCCL_MULTI_GPU=1: `world` ranks, one GPU each): every
rank steps on its own 2 images of a 2 * world-image batch over RCCL and saves
its parameter arena to $LD_RCCL_OUT/rank{r}.npy."""
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
    multi = os.environ.get('LD_RCCL_MULTI_GPU') == '1'
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) if multi else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            **(dict(device_id=dev) if multi else {}))
    from test_gpu_rccl import _batch, _one_step
    data = _batch(2 * world, 33, dev, lo=2 * rank, hi=2 * rank + 2)
    tr, _ = _one_step(dev, data, steps=1)
    np.save(os.path.join(os.environ['LD_RCCL_OUT'], f'rank{rank}.npy'),
            tr.arena.flat_param.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
