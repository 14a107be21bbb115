"""The LD step's cross-rank semantics on 2 gloo ranks with REAL loss-block data
(VERDICT r3 next #9), against the reference executed under a real 2-rank gloo
group (tests/golden/lossblock_2rank.npz, oracle/gen_golden.py
gen_lossblock_2rank):

  rank r holds lossblock case r (different GT counts, different images);
  its normaliser partials (sum_img max(P_img, 1); sum weight_targets + 1e-6) go
  through ld_amd.heads.GFLHead._norm_reducer -- ONE packed all-reduce where the
  reference does two reduce_mean(...).item() (ld_head.py:338-341, 362-365) --,
  the loss table of the rank is formed with the reduced normalisers, and
  ld_amd.detectors' _parse_losses reduces the 9 logged values with ONE packed
  all-reduce where the reference does one per key (base.py:211-216).

The arithmetic around the collectives is the CPU oracle's (the HIP loss block
cannot run here); the collectives, the LossDict and _parse_losses are the
product's own code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tag, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import ld_oracle as O
        from ld_amd import synthetic
        from ld_amd.detectors import SingleStageDetector
        from ld_amd.heads import GFLHead, LDHead
        g = np.load(os.path.join(HERE, 'golden', 'lossblock.npz'))
        g2 = np.load(os.path.join(HERE, 'golden', 'lossblock_2rank.npz'))
        name = str(g2[tag + '_cases'][rank])
        cfg = g[name + '_cfg']
        pad, img_shape = tuple(cfg[:2]), tuple(cfg[2:4])
        bseed, hseed = int(cfg[4]), int(cfg[5])
        num_gt = [int(x) for x in g[name + '_num_gt']]
        batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed)
        hi = {k: [t.numpy() for t in v] for k, v in hi.items()}
        t = O.get_targets(sizes, batch['img_metas'],
                          [b.numpy() for b in batch['gt_bboxes']],
                          [l.numpy() for l in batch['gt_labels']])
        # this rank's normaliser partials (what the HIP prepass leaves in norm[0:2])
        local = O.ld_loss_block(hi['cls'], hi['reg'], hi['t_cls'], hi['t_reg'],
                                hi['x'], hi['t_x'], t, with_grad=False)
        norm = torch.tensor([float(t['num_total_pos']), local['avg_factor'], 0.0, 0.0])
        mine = norm.clone()
        GFLHead._norm_reducer()(norm)  # the product's ONE packed all-reduce
        both = [torch.empty(4) for _ in range(world)]
        dist.all_gather(both, mine)
        want = torch.stack(both).mean(0)
        assert torch.allclose(norm, want, rtol=1e-7, atol=0)
        it = iter([float(norm[0]), float(norm[1])])
        out = O.ld_loss_block(hi['cls'], hi['reg'], hi['t_cls'], hi['t_reg'],
                              hi['x'], hi['t_x'], t, with_grad=False,
                              reduce_mean=lambda v: next(it))
        # ... equals the table the reference computed on this rank of ITS group
        ref = g2[f'{tag}_r{rank}_losses']
        np.testing.assert_allclose(out['losses'], ref, rtol=2e-5, atol=2e-6)
        assert out['num_total_samples'] == max(float(want[0]), 1.0)
        # and differs from the single-process table wherever a normaliser enters
        single = g[name + '_losses']
        assert not np.allclose(ref[0], single[0], rtol=1e-3), 'QFL row unchanged?'
        # the product's LossDict + _parse_losses (ONE packed all-reduce)
        table = torch.from_numpy(out['losses'].astype(np.float32))
        losses = LDHead._loss_dict(None, table)
        assert list(losses.keys()) == LOSS_KEYS
        loss, log_vars = SingleStageDetector._parse_losses(None, losses)
        np.testing.assert_allclose(float(loss), float(g2[f'{tag}_r{rank}_loss']),
                                   rtol=2e-5)
        for k, r in zip(LOSS_KEYS + ['loss'], g2[f'{tag}_r{rank}_log_vars']):
            np.testing.assert_allclose(log_vars[k], r, rtol=2e-5, atol=2e-6, err_msg=k)
        ret[rank] = 'ok'
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('tag', ['small_pair', 'c2_pair'])
def test_ld_cross_rank_normalisers_and_log_vars(tag):
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, 'worker failed'
    assert dict(ret) == {0: 'ok', 1: 'ok'}
