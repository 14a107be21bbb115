"""GPU end-to-end parity (-m gpu): the whole LD train step through the mmdet
API mirror on the MI355X -- student+teacher dual forward on the HIP conv/norm
stack, batched targets, fused loss block, backward -- against
 (1) golden loss tables / gradient norms produced by the reference itself with
     the same seeded state_dicts and synthetic batch, and
 (2) the CPU oracle (torch-CPU nets + numpy loss block) on the same inputs.
Tolerance: fp32 losses within 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CASES = [  # name, student depth, loss_im weight
    ('small_r50', 50, 2.0),
    ('c1_r18', 18, 0.0),
    ('c2_r50', 50, 2.0),
]
if os.environ.get('LD_TEST_FULL') == '1':  # every golden case (slower)
    CASES.insert(0, ('tiny_r18', 18, 0.0))
LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im']


def _setup(golden, name, sdepth, lw_im):
    from ld_amd import model_zoo, synthetic
    dev = torch.device('cuda:0')
    g = golden['e2e']
    cfg = g[name + '_cfg']
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
    num_gt = [int(x) for x in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt,
                                      bseed)
    det = model_zoo.build_seeded_ld_detector(sdepth, 101, dev,
                                             loss_im_weight=lw_im)
    dbatch = dict(img=batch['img'].to(dev), img_metas=batch['img_metas'],
                  gt_bboxes=[b.to(dev) for b in batch['gt_bboxes']],
                  gt_labels=[l.to(dev) for l in batch['gt_labels']])
    return g, det, batch, dbatch


@pytest.mark.parametrize('name,sdepth,lw_im', CASES,
                         ids=[c[0] for c in CASES])
def test_train_step_vs_reference_golden(golden, name, sdepth, lw_im):
    g, det, batch, dbatch = _setup(golden, name, sdepth, lw_im)
    losses = det(**dbatch)
    assert list(losses.keys()) == LOSS_KEYS
    table = torch.stack([torch.stack(losses[k]) for k in LOSS_KEYS])
    loss, log_vars = det._parse_losses(losses)
    loss.backward()
    torch.cuda.synchronize()
    got = table.detach().cpu().numpy().astype(np.float64)
    ref = g[name + '_losses']
    err = np.abs(got - ref)
    print(name, 'max abs loss err', err.max(), '\n', got, '\n', ref)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    ref_log = g[name + '_log_vars']
    for k, r in zip(LOSS_KEYS + ['loss'], ref_log):
        np.testing.assert_allclose(log_vars[k], r, rtol=1e-4, atol=1e-4,
                                   err_msg=k)
    # gradient fingerprints of every trainable parameter
    names = [str(k) for k in g[name + '_grad_names']]
    norms = g[name + '_grad_norms']
    params = dict(det.named_parameters())
    bad = []
    for k, r in zip(names, norms):
        assert params[k].grad is not None, k
        got_n = float(params[k].grad.double().norm())
        if not np.isclose(got_n, r, rtol=1e-3, atol=1e-6):
            bad.append((k, got_n, r))
    assert not bad, f'{len(bad)} grad norms off, first: {bad[:5]}'
    # ... and two pseudo-random projections of every gradient (sign / order /
    # layout sensitive).  A gradient with relative L2 error e moves a random
    # projection by ~ e * |g| * |probe| / sqrt(n); e <= 5e-3 is the element-wise
    # bound of test_features_vs_cpu_oracle, asserted here at 4 sigma.
    if name + '_grad_proj' in g.files:
        from ld_amd import synthetic
        proj = g[name + '_grad_proj']
        off = []
        for k, rn, pr in zip(names, norms, proj):
            gflat = params[k].grad.double().reshape(-1).cpu().numpy()
            for sd in (0, 1):
                probe = synthetic.grad_probe(gflat.size, sd)
                got_p = float(gflat @ probe)
                tol = 4 * 5e-3 * rn * np.linalg.norm(probe) / \
                    np.sqrt(gflat.size) + 1e-7
                if abs(got_p - pr[sd]) > tol:
                    off.append((k, sd, got_p, float(pr[sd]), tol))
        assert not off, f'{len(off)} gradient projections off: {off[:5]}'
    # frozen parameters received nothing
    for k, p in params.items():
        if not p.requires_grad:
            assert p.grad is None, k


@pytest.mark.parametrize('name,sdepth', [('small_r50', 50), ('c2_r50', 50)])
def test_train_step_gradient_elements_vs_reference(golden, name, sdepth):
    """Whole-step parameter gradients ELEMENT-WISE, at the
    small and the BASELINE config-2 size: (1) 256 sampled elements of every
    trainable parameter against the reference's own gradients, 1e-3 |ref| +
    1e-3 max|g| per element; (2) against the float64 evaluation of the same
    step: our error is within 3 x the reference's own fp32 error; (3) every
    gradient norm within 1e-3 of the reference's.  tests/_gradcheck.py states
    why 1e-5 is not a bar fp32 can meet on this net (the reference misses the
    float64 gradients by up to 4e-4 of max|g|)."""
    from _gradcheck import check_grad_samples, check_grad_truth64
    g, det, batch, dbatch = _setup(golden, name, sdepth, 2.0)
    losses = det(**dbatch)
    loss, _ = det._parse_losses(losses)
    loss.backward()
    torch.cuda.synchronize()
    params = dict(det.named_parameters())
    worst = check_grad_samples(golden, name, params)
    print(name, 'vs reference, max|err|/max|g| top 5:',
          [(f'{r:.1e}', k) for r, k in worst[:5]],
          'median', f'{np.median([r for r, _ in worst]):.1e}')
    rows = check_grad_truth64(golden, name, params)
    print(name, 'vs float64 (ours, reference) top 5:',
          [(f'{a:.1e}', f'{b:.1e}', k) for a, b, k in rows[:5]], 'median ours',
          f'{np.median([a for a, _, _ in rows]):.1e}', 'median reference',
          f'{np.median([b for _, b, _ in rows]):.1e}')
    names = [str(k) for k in g[name + '_grad_names']]
    bad = []
    for k, r in zip(names, g[name + '_grad_norms']):
        got_n = float(params[k].grad.double().norm())
        if not np.isclose(got_n, r, rtol=1e-3, atol=1e-7):
            bad.append((k, got_n, r))
    assert not bad, f'{len(bad)} grad norms off at 1e-3, first: {bad[:5]}'


def test_features_vs_cpu_oracle(golden):
    """Intermediate tensors (FPN features, head outputs) and parameter
    gradients element-wise against the torch-CPU oracle at a small size."""
    import net_oracle as NO
    from ld_amd import synthetic
    g, det, batch, dbatch = _setup(golden, 'small_r50', 50, 2.0)
    ssd = {k: v.detach().cpu() for k, v in det.state_dict().items()}
    tsd = {k: v.detach().cpu()
           for k, v in det.teacher_model.state_dict().items()}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref = NO.ld_train_step(ssd, tsd, batch, 50, 101)
    x = det.extract_feat(dbatch['img'])
    cls, reg = det.bbox_head(x)
    for l in range(5):
        for nm, a, b in (('feat', x[l], ref['feats'][l]),
                         ('cls', cls[l], ref['cls'][l]),
                         ('reg', reg[l], ref['reg'][l])):
            a, b = a.detach().cpu().double(), b.detach().double()
            scale = float(b.abs().max()) + 1e-12
            err = float((a - b).abs().max())
            assert err <= 2e-4 * scale + 1e-6, f'{nm}[{l}] err {err} scale {scale}'
    losses = det(**dbatch)
    loss, _ = det._parse_losses(losses)
    loss.backward()
    params = dict(det.named_parameters())
    rels = {}
    for k, gref in ref['grads'].items():
        a = params[k].grad.detach().cpu().double()
        b = gref.double()
        rels[k] = float((a - b).norm() / (b.norm() + 1e-12))
    bad = sorted(((v, k) for k, v in rels.items() if not v < 5e-3), reverse=True)
    print('worst relative grad error', max(rels.values()))
    assert not bad, f'{len(bad)} of {len(rels)} gradients off: {bad[:8]}'


def test_arena_direct_grads_match_autograd(golden):
    """With a GradArena the backward kernels accumulate straight into the flat
    gradient arena (no temporaries, no add kernels) and autograd sees None for
    those inputs; the result must be the very same bits as plain autograd
    accumulation, and every parameter must be reported ready exactly once."""
    from ld_amd.train import GradArena
    g, det, batch, dbatch = _setup(golden, 'tiny_r18', 18, 0.0)
    losses = det(**dbatch)
    loss, _ = det._parse_losses(losses)
    loss.backward()
    ref = {k: p.grad.detach().clone() for k, p in det.named_parameters()
           if p.grad is not None}
    for p in det.parameters():
        p.grad = None
    arena = GradArena(list(det.parameters()))
    arena.zero_grad()
    losses = det(**dbatch)
    loss, _ = det._parse_losses(losses)
    loss.backward()
    torch.cuda.synchronize()
    assert len(ref) == len(arena.params)
    for k, p in det.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad.data_ptr() == p._ld_grad.data_ptr(), k
        assert torch.equal(p.grad, ref[k]), \
            f'{k}: max diff {(p.grad - ref[k]).abs().max().item()}'
    assert arena._seen == {id(p) for p in arena.params}
    assert arena._ready == [bk['n'] for bk in arena.buckets]


def test_trainer_two_steps_reduce_loss(golden):
    """SGDTrainer (flat arenas, hooks, fused SGD): two steps on the same batch
    run, keep gradients in the arena, and change the weights."""
    from ld_amd.train import SGDTrainer
    g, det, batch, dbatch = _setup(golden, 'tiny_r18', 18, 0.0)
    tr = SGDTrainer(det, lr=0.0025, momentum=0.9, weight_decay=1e-4)
    p0 = tr.arena.flat_param.clone()
    out1 = tr.step(dbatch)
    out2 = tr.step(dbatch)
    torch.cuda.synchronize()
    l1, l2 = float(out1['loss']), float(out2['loss'])
    assert np.isfinite(l1) and np.isfinite(l2)
    np.testing.assert_allclose(l1, g['tiny_r18_log_vars'][-1], rtol=1e-4)
    assert not torch.equal(p0, tr.arena.flat_param)
    assert float(tr.flat_momentum.abs().sum()) > 0
    for p in det.parameters():
        if p.requires_grad:
            assert p.grad.data_ptr() >= tr.arena.flat_grad.data_ptr()
    assert out2['log_vars']['loss'] == pytest.approx(l2, rel=1e-6)
    # the one-launch refresh after the optimizer step left every cached GEMM
    # weight image and BN coefficient equal to a fresh per-tensor computation
    import ctypes as C
    from ld_amd import layers as Y, lib as L
    lib = L.get_lib()
    st = L.stream_ptr(dbatch['img'].device)
    n_w = n_b = 0
    for ref in Y._WT_REG.values():
        w = ref()
        if w is None or getattr(w, '_ld_images', None) is None:
            continue
        cache = w._ld_images
        if cache['stamp'] != (w._version, Y._PARAM_GEN[0], w.data_ptr(), False):
            continue
        cout, cin, kh, kw = w.shape
        fwd = torch.empty_like(cache['fwd'])
        bwd = torch.empty_like(cache['bwd']) if cache['bwd'] is not None \
            else None
        L.check(lib.ld_conv_weight_transform(L.ptr(w), cout, cin, kh, kw,
                                             L.ptr(fwd), L.ptr(bwd), st), 'wt')
        assert torch.equal(fwd, cache['fwd'])
        if bwd is not None and cache['bwd_stamp'] == cache['stamp']:
            assert torch.equal(bwd, cache['bwd'])
        n_w += 1
    for ref in Y._BN_REG.values():
        gmm = ref()
        if gmm is None or gmm._ld_bn[0][-1] != Y._PARAM_GEN[0]:
            continue
        beta, mean, var, eps = gmm._ld_bn_src
        fresh = Y._bn_prepare(gmm, beta, mean, var, eps)
        for a, b in zip(fresh, gmm._ld_bn[1]):
            assert torch.equal(a, b)
        n_b += 1
    assert n_w >= 15 and n_b >= 15, (n_w, n_b)
