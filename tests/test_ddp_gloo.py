"""The multi-GPU path on CPU: world_size-2 gloo processes.  Checks that
 (i) GradArena's bucketed hook-driven all-reduce produces the gradients a
     single process gets on the concatenated batch,
 (ii) the loss normalisers are averaged across ranks the way the reference's
     reduce_mean does (ld_head.py:338-341,362-365),
 (iii) _parse_losses' single packed all-reduce equals per-key means, and
 (iv) ranks that were initialised with DIFFERENT seeds hold rank 0's
     parameters, frozen parameters and buffers after GradArena's constructor
     (the broadcast MMDistributedDataParallel does, apis/train.py:74-84), and
 (v) a step that only one rank performs under suspend_collectives (the warm-up
     of a lazily captured hipGraph) issues no collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from ld_amd.train import GradArena
        from ld_amd.heads import GFLHead
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(),
                                    torch.nn.Linear(32, 8),
                                    torch.nn.Linear(8, 4))
        full_x = torch.randn(8, 16)
        full_y = torch.randn(8, 4)
        # single-process reference on the whole batch (mean over 8 samples)
        ref_model = torch.nn.Sequential(torch.nn.Linear(16, 32),
                                        torch.nn.Tanh(),
                                        torch.nn.Linear(32, 8),
                                        torch.nn.Linear(8, 4))
        ref_model.load_state_dict(model.state_dict())
        ((ref_model(full_x) - full_y)**2).sum(1).mean().backward()
        ref = [p.grad.clone() for p in ref_model.parameters()]
        # tiny buckets -> several collectives, fired from the hooks
        arena = GradArena(list(model.parameters()), bucket_bytes=600)
        assert len(arena.buckets) >= 3
        for step in range(2):  # second step: arena re-zeroed correctly
            arena.zero_grad()
            sl = slice(rank * 4, rank * 4 + 4)
            ((model(full_x[sl]) - full_y[sl])**2).sum(1).mean().backward()
            arena.finish()
            for p, r in zip(model.parameters(), ref):
                got = p.grad / world  # SGD kernel folds 1/world in
                assert torch.allclose(got, r, rtol=1e-5, atol=1e-6), step
        # (v) a step only ONE rank performs (the warm-up of a lazily captured
        # hipGraph, AutoStepper): under suspend_collectives it issues no
        # collective, so the peer -- which does nothing meanwhile -- is not left
        # with an unmatched all-reduce, and the next real step still reduces
        from ld_amd.train import collectives_on, suspend_collectives
        if rank == 0:
            with suspend_collectives():
                assert not collectives_on()
                arena.zero_grad()
                ((model(full_x[:4]) - full_y[:4])**2).sum(1).mean().backward()
                arena.finish()
                local = [p.grad.clone() for p in model.parameters()]
            lm = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(),
                                     torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
            lm.load_state_dict(model.state_dict())
            ((lm(full_x[:4]) - full_y[:4])**2).sum(1).mean().backward()
            for a, b in zip(local, lm.parameters()):
                assert torch.allclose(a, b.grad, rtol=1e-5, atol=1e-6)
        assert collectives_on()
        arena.zero_grad()
        sl = slice(rank * 4, rank * 4 + 4)
        ((model(full_x[sl]) - full_y[sl])**2).sum(1).mean().backward()
        arena.finish()
        for p, r in zip(model.parameters(), ref):
            assert torch.allclose(p.grad / world, r, rtol=1e-5, atol=1e-6)
        # (ii) normaliser reduction
        red = GFLHead._norm_reducer()
        norm = torch.tensor([3.0 + rank, 1.5 * (rank + 1), 0., 0.])
        red(norm)
        assert torch.allclose(norm[:2], torch.tensor([3.5, 2.25]))
        # (iii) packed log-var reduction
        from ld_amd.detectors import SingleStageDetector
        losses = dict(loss_a=[torch.tensor(1.0 + rank), torch.tensor(2.0)],
                      loss_b=torch.tensor([2.0 * rank, 4.0]),
                      acc=torch.tensor(10.0 * rank))
        loss, lv = SingleStageDetector._parse_losses(None, losses)
        assert float(loss) == (3.0 + rank) + (rank + 2.0)
        assert abs(lv['loss_a'] - 3.5) < 1e-6 and abs(lv['loss_b'] - 2.5) < 1e-6
        assert abs(lv['acc'] - 5.0) < 1e-6 and abs(lv['loss'] - 6.0) < 1e-6
        # (iv) different per-rank initialisation -> rank 0's state everywhere
        torch.manual_seed(100 + rank)
        m2 = torch.nn.Sequential(torch.nn.Linear(6, 5),
                                 torch.nn.BatchNorm1d(5),
                                 torch.nn.Linear(5, 3))
        m2[1].running_mean.normal_()
        m2[1].num_batches_tracked.fill_(7 + rank)
        m2[2].weight.requires_grad_(False)  # a frozen parameter
        frozen = [p for p in m2.parameters() if not p.requires_grad]
        arena2 = GradArena(list(m2.parameters()),
                           extra_state=frozen + list(m2.buffers()))
        state = torch.cat([arena2.flat_param, m2[2].weight.reshape(-1),
                           m2[1].running_mean,
                           m2[1].num_batches_tracked.reshape(1).float()])
        got = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(got, state)
        assert all(torch.equal(g, got[0]) for g in got)
        torch.manual_seed(100)  # what rank 0 drew
        ref2 = torch.nn.Linear(6, 5)
        assert torch.equal(m2[0].weight, ref2.weight)
        assert int(m2[1].num_batches_tracked) == 7
        assert m2[0].weight.data_ptr() >= arena2.flat_param.data_ptr()
        ret[rank] = 'ok'
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, 'worker failed'
    assert dict(ret) == {0: 'ok', 1: 'ok'}
