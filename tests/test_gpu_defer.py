"""GPU tests (-m gpu) of round 5's per-bucket finalisation of parameter gradients:
ld_conv_wgrad_partial + ld_wgrad_reduce_batch and the deferring form of the BN
backward + ld_bn_bwd_finalize_batch must give the per-layer launches' results BIT
FOR BIT (the same additions in the same order), alone and inside a whole step."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # name, N, Cin, Cout, k, stride, pad, levels
    ('1x1_64_256', 2, 64, 256, 1, 1, 0, ((20, 34), )),
    ('1x1_s2_256_512', 2, 256, 512, 1, 2, 0, ((20, 34), )),
    ('3x3_64_64', 2, 64, 64, 3, 1, 1, ((24, 40), )),
    ('3x3_s2_128', 2, 128, 128, 3, 2, 1, ((26, 42), )),
    ('3x3_256_256_levels', 2, 256, 256, 3, 1, 1,
     ((20, 28), (10, 14), (5, 7), (3, 4), (2, 2))),
    ('3x3_256_80_levels', 2, 256, 80, 3, 1, 1,
     ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))),
    ('3x3_256_68_levels', 1, 256, 68, 3, 1, 1,
     ((12, 20), (6, 10), (3, 5), (2, 3), (1, 2))),
    ('3x3_3_64_stemlike', 1, 16, 48, 3, 1, 1, ((30, 30), )),
    ('1x1_1024_256', 2, 1024, 256, 1, 1, 0, ((13, 21), )),
]


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('family', [0, 1, 2])
def test_wgrad_partial_plus_batch_equals_per_layer_launch(family):
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    st = L.stream_ptr(dev)
    jobs, blocks, want, outs, keep = [], [], [], [], []
    for name, N, cin, cout, k, stride, pad, levels in CASES:
        if family == 2 and (cin % 32 or cout % 32):
            continue
        g = torch.Generator().manual_seed(cin * 3 + cout)
        d, _ = Y.conv_desc(N, cin, cout, k, k, stride, pad, levels)
        x = torch.randn(N, cin, d.Pin, generator=g).to(dev)
        dy = torch.randn(N, cout, d.Pout, generator=g).to(dev)
        base = torch.randn(cout, cin, k, k, generator=g).to(dev)
        need = lib.ld_conv_wgrad_workspace_bytes(C.byref(d))
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        if family == 2:
            xin, dyin = Y.to_c8(x), Y.to_c8(dy)
            ref_fn = lib.ld_conv_bf16_wgrad_c8
        else:
            xin, dyin = x, dy
            ref_fn = lib.ld_conv_bf16_wgrad if family == 1 else lib.ld_conv_wgrad
        for acc in (0, 1):
            ref = base.clone()
            L.check(ref_fn(C.byref(d), L.ptr(xin), L.ptr(dyin), L.ptr(ref), acc,
                           L.ptr(ws), ws.numel(), st), 'per-layer wgrad')
            want.append(ref)
            slabs = torch.full((need, ), 0xA5, dtype=torch.uint8, device=dev)
            job = L.WgradJobT()
            L.check(lib.ld_conv_wgrad_partial(C.byref(d), family, L.ptr(xin), L.ptr(dyin),
                                              L.ptr(slabs), slabs.numel(), C.byref(job),
                                              st), 'ld_conv_wgrad_partial')
            assert job.splits >= 1 and job.Cout == cout and job.Cin == cin and \
                job.ntaps == k * k and job.slabs == slabs.data_ptr()
            out = base.clone()
            job.dw, job.accumulate = out.data_ptr(), acc
            jobs.append(job)
            blocks.append((k * k * cout * cin + 1023) // 1024)
            outs.append(out)
            keep.append((slabs, xin, dyin))
    tab, bmap, nb = Y._job_table(jobs, blocks, dev)
    L.check(lib.ld_wgrad_reduce_batch(L.ptr(tab), L.ptr(bmap), nb, st), 'batch')
    torch.cuda.synchronize()
    assert len(outs) >= 8
    for i, (o, w) in enumerate(zip(outs, want)):
        assert torch.isfinite(o).all()
        assert torch.equal(o, w), (family, i, float((o - w).abs().max()))


@pytest.mark.parametrize('c8', [False, True])
def test_bn_backward_deferred_equals_per_layer_finalize(c8):
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(4)
    jobs, blocks, want, got, keep = [], [], [], [], []
    for (N, c, P) in ((2, 64, 1600), (2, 256, 4200), (1, 512, 1052), (2, 128, 16800)):
        dy, y, x = (torch.randn(N, c, P, generator=g).to(dev) for _ in range(3))
        scale, mean = (torch.randn(c, generator=g).to(dev) for _ in range(2))
        rstd = torch.rand(c, generator=g).to(dev) + 0.5
        need = lib.ld_bn_act_backward_workspace_bytes(N, c, P)
        for acc in (0, 1):
            base_g, base_b = (torch.randn(c, generator=g).to(dev) for _ in range(2))
            outs = []
            for mode in (acc, L.LD_GRAD_DEFER):
                ws = torch.zeros(need, dtype=torch.uint8, device=dev)
                dx = torch.empty_like(x)
                dg, db = base_g.clone(), base_b.clone()
                if c8:
                    dx8 = torch.empty(N * c * P, dtype=torch.bfloat16, device=dev)
                    rc = lib.ld_bn_act_backward_c8(
                        L.ptr(dy), L.ptr(y), L.ptr(x), L.ptr(scale), L.ptr(mean),
                        L.ptr(rstd), N, c, P, 1, L.ptr(dx), L.ptr(dx8), None, L.ptr(dg),
                        L.ptr(db), mode, L.ptr(ws), ws.numel(), st)
                else:
                    rc = lib.ld_bn_act_backward(
                        L.ptr(dy), L.ptr(y), L.ptr(x), L.ptr(scale), L.ptr(mean),
                        L.ptr(rstd), N, c, P, 1, L.ptr(dx), None, L.ptr(dg), L.ptr(db),
                        mode, L.ptr(ws), ws.numel(), st)
                L.check(rc, 'bn backward')
                outs.append((dx, dg, db, ws))
            (dx0, dg0, db0, _), (dx1, dg1, db1, ws1) = outs
            torch.cuda.synchronize()
            assert torch.equal(dx0, dx1)
            # the deferring call leaves the parameter gradients alone
            assert torch.equal(dg1, base_g) and torch.equal(db1, base_b)
            job = L.BnFinJobT()
            job.partial, job.dgamma, job.dbeta = ws1.data_ptr(), dg1.data_ptr(), db1.data_ptr()
            job.C, job.accumulate = c, acc
            job.nsplit = lib.ld_bn_act_backward_nsplit(N, c, P, 1 if c8 else 0)
            assert 1 <= job.nsplit <= 256
            jobs.append(job)
            blocks.append((c + 15) // 16)
            want.append((dg0, db0))
            got.append((dg1, db1))
            keep.append(ws1)
    from ld_amd import layers as Y
    tab, bmap, nb = Y._job_table(jobs, blocks, dev)
    L.check(lib.ld_bn_bwd_finalize_batch(L.ptr(tab), L.ptr(bmap), nb, st), 'batch')
    torch.cuda.synchronize()
    for (a0, b0), (a1, b1) in zip(want, got):
        assert torch.equal(a0, a1) and torch.equal(b0, b1)


def _step(defer, precision, steps=2):
    from ld_amd import layers as Y
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import SGDTrainer
    dev = _dev()
    prev_d, prev_p = Y._DEFER_ON[0], Y.get_precision()
    Y._DEFER_ON[0] = defer
    Y.set_precision(precision)
    try:
        det = model_zoo.build_seeded_ld_detector(50, 101, dev)
        tr = SGDTrainer(det, lr=0.0, bucket_bytes=8 << 20)  # lr 0: both steps see the same weights
        b = synthetic.synthetic_batch(2, (256, 320), (256, 320), [4, 2], 31)
        d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                 gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                 gt_labels=[x.to(dev) for x in b['gt_labels']])
        before = dict(Y.DEFER_STATS)
        for _ in range(steps):
            out = tr.step(d)
        torch.cuda.synchronize()
        assert not Y.deferred_pending()
        stats = {k: Y.DEFER_STATS[k] - before[k] for k in before}
        names = {id(p): k for k, p in det.named_parameters()}
        spans = [(names[id(p)], o, p.numel())
                 for p, o in zip(tr.arena.order, tr.arena.offsets)]
        return (tr.arena.flat_grad.clone(), tr.arena.flat_param.clone(),
                float(out['log_vars']['loss']), stats, len(tr.arena.buckets), spans)
    finally:
        Y._DEFER_ON[0] = prev_d
        Y.set_precision(prev_p)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_train_step_deferred_gradients_bit_identical(precision):
    g0, p0, l0, s0, _, _ = _step(False, precision)
    g1, p1, l1, s1, nb, spans = _step(True, precision)
    assert s0['wgrad_jobs'] == 0 and s0['flushes'] == 0
    # 53 trainable convs and 42 trainable norms in the R50 student, two steps
    assert s1['wgrad_jobs'] >= 2 * 50 and s1['bn_jobs'] >= 2 * 40, s1
    assert nb >= 3 and s1['flushes'] <= 2 * (nb + 1), (s1, nb)
    assert s1['bn_jobs'] >= 2 * 50  # 42 norms + 10 conv biases per step
    # the job tables are uploaded in the first step only
    assert s1['tables_built'] <= 2 * (nb + 1), s1
    assert l0 == l1
    # conv BIASES (neck convs, gfl_cls / gfl_reg) take the split partial sums when
    # deferred and the one-block-per-channel kernel otherwise: a different order of
    # fp64 additions (1e-7 relative at most); everything else bit for bit
    nbias = 0
    for name, o, n in spans:
        a, b = g0[o:o + n], g1[o:o + n]
        if name.endswith('conv.bias') or name.endswith('gfl_cls.bias') or \
                name.endswith('gfl_reg.bias'):
            nbias += 1
            assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-12, name
        else:
            assert torch.equal(a, b), name
    assert nbias == 10


def test_bias_grad_partial_equals_one_block_kernel():
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = _dev()
    lib = L.get_lib()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(8)
    jobs, blocks, outs, want, keep = [], [], [], [], []
    for (N, c, P) in ((2, 256, 16800), (2, 80, 22400), (1, 68, 1053), (2, 256, 77)):
        dy = torch.randn(N, c, P, generator=g).to(dev)
        ref = torch.empty(c, device=dev)
        L.check(lib.ld_bias_grad(L.ptr(dy), N, c, P, L.ptr(ref), 0, st), 'bias')
        ns = lib.ld_bias_grad_nsplit(N, c, P)
        part = torch.full((c * ns * 2, ), float('nan'), dtype=torch.float64, device=dev)
        L.check(lib.ld_bias_grad_partial(L.ptr(dy), N, c, P, L.ptr(part),
                                         part.numel() * 8, st), 'partial')
        out = torch.zeros(c, device=dev)
        job = L.BnFinJobT()
        job.partial, job.dgamma, job.dbeta = part.data_ptr(), None, out.data_ptr()
        job.C, job.nsplit, job.accumulate = c, ns, 0
        jobs.append(job)
        blocks.append((c + 15) // 16)
        outs.append(out)
        want.append((ref, dy.double().sum((0, 2))))
        keep.append((part, dy))
    tab, bmap, nb = Y._job_table(jobs, blocks, dev)
    L.check(lib.ld_bn_bwd_finalize_batch(L.ptr(tab), L.ptr(bmap), nb, st), 'batch')
    torch.cuda.synchronize()
    for o, (ref, exact) in zip(outs, want):
        scale = float(exact.abs().max())
        assert float((o.double() - exact).abs().max()) <= 2e-7 * scale + 1e-9
        assert float((o - ref).abs().max()) <= 4e-7 * scale + 1e-9
