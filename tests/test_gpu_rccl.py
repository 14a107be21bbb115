"""The RCCL path on the one GPU the box has (-m gpu; SURVEY.md section 8e "Test
without 8 GPUs", reference recipe mmdet/apis/train.py:74-127).

 (1) world_size = 1 ``nccl`` (= RCCL) group + LD_FORCE_COLLECTIVES=1: every
     collective of the train step is issued for real -- constructor broadcast,
     the bucketed gradient all-reduces fired from the backward hooks, the
     packed loss-normaliser all-reduce inside the loss block, the packed log
     all-reduce -- and one ``SGDTrainer.step`` must leave the parameter arena
     BIT-IDENTICAL to the step without collectives (a 1-rank sum is the
     identity, the 1/world scaling is a multiplication by 1.0).
 (2) two ranks sharing cuda:0, if this RCCL build accepts a duplicate device
     (skipped with the reason otherwise): 2 ranks x 2 images must give the
     parameters one process gets on the 4-image batch, and both ranks must
     hold identical arenas afterwards.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(n_img, seed, dev, lo=0, hi=None):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(n_img, (128, 150), (128, 160),
                                  ([3, 2, 4, 1] * 4)[:n_img], seed)
    hi = n_img if hi is None else hi
    return dict(img=b['img'][lo:hi].to(dev), img_metas=b['img_metas'][lo:hi],
                gt_bboxes=[x.to(dev) for x in b['gt_bboxes'][lo:hi]],
                gt_labels=[x.to(dev) for x in b['gt_labels'][lo:hi]])


def _one_step(dev, data, steps=2):
    from ld_amd import model_zoo
    from ld_amd.train import SGDTrainer
    det = model_zoo.build_seeded_ld_detector(18, 18, dev, loss_im_weight=2.0)
    tr = SGDTrainer(det, lr=0.01, bucket_bytes=4 << 20)
    out = None
    for _ in range(steps):
        out = tr.step(data)
    torch.cuda.synchronize()
    return tr, out


def test_forced_collectives_one_rank_bit_exact(monkeypatch):
    import torch.distributed as dist
    dev = torch.device('cuda:0')
    data = _batch(2, 21, dev)
    monkeypatch.delenv('LD_FORCE_COLLECTIVES', raising=False)
    assert not dist.is_initialized()
    tr0, out0 = _one_step(dev, data)
    ref_param = tr0.arena.flat_param.clone()
    ref_mom = tr0.flat_momentum.clone()
    ref_logs = dict(out0['log_vars'])

    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('LD_FORCE_COLLECTIVES', '1')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        from ld_amd import train as T
        assert T.collectives_on()
        # count what is issued: the test must not pass because nothing ran
        calls = dict(all_reduce=0, broadcast=0)
        real_ar, real_bc = dist.all_reduce, dist.broadcast

        def _ar(*a, **k):
            calls['all_reduce'] += 1
            return real_ar(*a, **k)

        def _bc(*a, **k):
            calls['broadcast'] += 1
            return real_bc(*a, **k)

        monkeypatch.setattr(dist, 'all_reduce', _ar)
        monkeypatch.setattr(dist, 'broadcast', _bc)
        tr1, out1 = _one_step(dev, data)
        nb = len(tr1.arena.buckets)
        assert nb >= 2, 'want several gradient buckets'
        # per step: nb gradient buckets + 1 normaliser pair + 1 packed logs
        assert calls['all_reduce'] == 2 * (nb + 2), calls
        assert calls['broadcast'] >= 2, calls  # arena + packed extra state
        assert torch.equal(tr1.arena.flat_param, ref_param)
        assert torch.equal(tr1.flat_momentum, ref_mom)
        logs = dict(out1['log_vars'])
        assert logs == ref_logs
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(
    os.environ.get('LD_TEST_RCCL_2RANK') != '1',
    reason='this RCCL build refuses two ranks on one device (measured: the '
           'launch fails with "duplicate GPU detected", profiles/'
           'r02_pytest_gpu_s1_bf16_rccl.txt); set LD_TEST_RCCL_2RANK=1 on a '
           'box where it is allowed')
def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    dev = torch.device('cuda:0')
    # single process, 4 images
    tr, _ = _one_step(dev, _batch(4, 33, dev), steps=1)
    ref = tr.arena.flat_param.detach().cpu().numpy()
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0', LD_RCCL_OUT=str(tmp_path))
    env.pop('LD_FORCE_COLLECTIVES', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port),
           os.path.join(REPO, 'tests', '_rccl_two_rank.py')]
    try:
        r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True,
                           text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip('2 ranks on one GPU: RCCL did not finish in 240 s')
    if r.returncode != 0:
        tail = (r.stderr or r.stdout)[-600:]
        pytest.skip('this RCCL build refuses two ranks on one device: ' + tail)
    got = [np.load(tmp_path / f'rank{i}.npy') for i in range(2)]
    assert np.array_equal(got[0], got[1]), 'ranks diverged'
    # fp32 reassociation between (2+2 averaged) and one 4-image batch
    np.testing.assert_allclose(got[0], ref, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize('world', [2, 8])
def test_multi_gpu_ranks_match_single_process(world, tmp_path):
    """`world` ranks, ONE GPU EACH (a multi-GPU node: skipped on the 1-GPU box),
    2 images per rank over RCCL / xGMI, against one process stepping on all 2 *
    world images: equal parameters on every rank, and equal to the single-process
    update within fp32 reassociation (the reference's MMDistributedDataParallel
    contract, mmdet/apis/train.py:74-82; cross-rank loss normalisers included:
    ld_head.py:338-341, 362-365)."""
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs on this node, '
                    f'{torch.cuda.device_count()} visible')
    dev = torch.device('cuda:0')
    tr, _ = _one_step(dev, _batch(2 * world, 33, dev), steps=1)
    ref = tr.arena.flat_param.detach().cpu().numpy()
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0', LD_RCCL_OUT=str(tmp_path),
               LD_RCCL_MULTI_GPU='1')
    env.pop('LD_FORCE_COLLECTIVES', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port),
           os.path.join(REPO, 'tests', '_rccl_two_rank.py')]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stderr or r.stdout)[-1500:]
    got = [np.load(tmp_path / f'rank{i}.npy') for i in range(world)]
    for g in got[1:]:
        assert np.array_equal(got[0], g), 'ranks diverged'
    np.testing.assert_allclose(got[0], ref, rtol=2e-4, atol=2e-6)
