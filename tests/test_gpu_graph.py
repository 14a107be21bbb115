"""GPU (-m gpu): the train step captured into a hipGraph (ld_amd.train.
GraphedStep) replays to the SAME bits as the eager step -- same kernels in the
same order, every launch entry point enqueue-only -- in fp32 and in bf16 mode,
and picks up new data copied into its static input buffers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(seed, dev):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(2, (128, 150), (128, 160), [3, 2], seed)
    return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                gt_labels=[x.to(dev) for x in b['gt_labels']])


def _trainer(dev):
    from ld_amd import model_zoo
    from ld_amd.train import SGDTrainer
    det = model_zoo.build_seeded_ld_detector(18, 18, dev, loss_im_weight=2.0)
    return SGDTrainer(det, lr=0.01)


@pytest.mark.parametrize('launcher', ['graph', 'list'])
@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_graphed_step_equals_eager(mode, launcher):
    """launcher='list': the captured graph re-issued by the C launch loop of
    csrc/graphlist.hip (ld_step_list_*) instead of hipGraphLaunch."""
    from ld_amd import layers as Y
    from ld_amd.train import GraphedStep
    dev = torch.device('cuda:0')
    Y.set_precision(mode)
    try:
        d1, d2 = _batch(21, dev), _batch(22, dev)
        eager = _trainer(dev)
        for d in (d1, d1, d1, d2):
            out_e = eager.step(d)
        torch.cuda.synchronize()
        tr = _trainer(dev)
        static = _batch(21, dev)
        g = GraphedStep(tr, static, warmup=2, launcher=launcher)  # two eager steps on d1
        if launcher == 'list':
            assert g.list.info['kernels'] > 100 and g.list.info['lanes'] >= 1
        out_g = g.replay()                     # third step on d1
        torch.cuda.synchronize()
        g.copy_inputs(d2)
        out_g = g.replay()                     # fourth step, new data
        torch.cuda.synchronize()
        assert torch.equal(tr.arena.flat_param, eager.arena.flat_param)
        assert torch.equal(tr.flat_momentum, eager.flat_momentum)
        assert dict(out_g['log_vars']) == dict(out_e['log_vars'])
        assert float(out_g['loss']) == float(out_e['loss'])
        assert np.isfinite(float(out_g['loss']))
    finally:
        Y.set_precision('fp32')


def _batch_g(seed, num_gt, dev, img_shape=(128, 150)):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(2, img_shape, (128, 160), num_gt, seed)
    return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                gt_labels=[x.to(dev) for x in b['gt_labels']])


def test_graphed_step_takes_real_batches():
    """VERDICT round 2, missing #5 + ADVICE: the captured step must serve what
    a COCO epoch hands it (kd_one_stage.py:52-65) -- a different number of GT
    boxes per image every iteration, a different pad_shape / img_shape, and a
    learning rate that follows its schedule.  Replays must equal the eager
    steps on the same sequence BIT FOR BIT."""
    from ld_amd.train import GraphedStep
    dev = torch.device('cuda:0')
    seq = [_batch_g(21, [3, 2], dev), _batch_g(21, [3, 2], dev),
           _batch_g(23, [6, 1], dev), _batch_g(24, [1, 9], dev, (120, 133)),
           _batch_g(25, [4, 4], dev)]
    # a smaller per-image pad_shape inside the same batch tensor: fewer valid
    # anchors (anchor_generator.py:293-300) -> different targets
    seq[3]['img_metas'][1]['pad_shape'] = (96, 128, 3)
    seq[4]['img_metas'][0]['pad_shape'] = (128, 96, 3)
    lrs = [0.01, 0.01, 0.01, 0.004, 0.02]
    eager = _trainer(dev)
    outs_e = []
    for d, lr in zip(seq, lrs):
        eager.lr = lr
        outs_e.append(eager.step(d))
    torch.cuda.synchronize()
    tr = _trainer(dev)
    g = GraphedStep(tr, _batch_g(21, [3, 2], dev), warmup=1, max_gt=16)
    assert g.dynamic_gt and g.static.max_gt == 16
    outs_g = []
    # step 0 ran as the warm-up; the capture itself executes nothing
    # no synchronisation between the replays: the host runs ahead of the GPU,
    # so the small host -> device updates (box counts, valid regions, lr) must
    # survive being queued behind a whole replay (lossblock.PinnedRing)
    for k, (d, lr) in enumerate(zip(seq[1:], lrs[1:])):
        tr.lr = lr
        g.copy_inputs(d)
        outs_g.append(g.replay()['loss'].clone())
    torch.cuda.synchronize()
    for k, l in enumerate(outs_g):
        assert float(l) == float(outs_e[1 + k]['loss']), k
    assert torch.equal(tr.arena.flat_param, eager.arena.flat_param)
    assert torch.equal(tr.flat_momentum, eager.flat_momentum)
    with pytest.raises(ValueError):
        g.copy_inputs(_batch_g(26, [17, 1], dev))  # more boxes than max_gt


@pytest.mark.parametrize('launcher', ['graph', 'list'])
@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_pipelined_graphed_step_equals_eager(mode, launcher):
    """train.PipelinedGraphedStep (two hipGraphs over two slots: student step
    of batch i + teacher forward of batch i + 1 in one replay, weight gradients
    on their side stream) against plain eager steps on the same batch
    sequence, different GT counts per batch: the same parameters bit for bit."""
    from ld_amd import layers as Y
    from ld_amd.train import PipelinedGraphedStep
    dev = torch.device('cuda:0')
    Y.set_precision(mode)
    try:
        b = [_batch_g(31, [3, 2], dev), _batch_g(32, [5, 1], dev),
             _batch_g(33, [2, 7], dev), _batch_g(34, [4, 4], dev),
             _batch_g(35, [1, 1], dev)]
        eager = _trainer(dev)
        seq = [b[0], b[1], b[0], b[1], b[2], b[3]]  # warm-up on both slots first
        for d in seq:
            out_e = eager.step(d)
        torch.cuda.synchronize()
        tr = _trainer(dev)
        ps = PipelinedGraphedStep(tr, b[0], b[1], warmup=1, max_gt=16,
                                  launcher=launcher)
        outs = [ps.step(b[1]), ps.step(b[2]), ps.step(b[3]), ps.step(b[4])]
        torch.cuda.synchronize()
        assert torch.equal(tr.arena.flat_param, eager.arena.flat_param)
        assert torch.equal(tr.flat_momentum, eager.flat_momentum)
        assert float(outs[-1]['loss']) == float(out_e['loss'])
    finally:
        Y.set_precision('fp32')


def test_teacher_prefetch_bit_identical():
    """SGDTrainer.step(data, next_data=...) runs the frozen teacher of the next
    batch under this step (KnowledgeDistillationSingleStageDetector.
    prefetch_teacher); three pipelined steps give exactly the parameters and
    losses of three plain steps, also when a DIFFERENT batch arrives than the
    one that was prefetched (the stale result is dropped)."""
    import torch
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import SGDTrainer
    dev = torch.device('cuda:0')

    def batch(seed):
        b = synthetic.synthetic_batch(2, (160, 200), (160, 224), 3, seed)
        return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                    gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                    gt_labels=[x.to(dev) for x in b['gt_labels']])

    batches = [batch(s) for s in (5, 6, 7, 8)]  # four DISTINCT tensors
    results = []
    hits = {}
    for mode in ('plain', 'pipelined', 'stale'):
        det = model_zoo.build_seeded_ld_detector(18, 18, dev)
        tr = SGDTrainer(det, lr=0.01)
        losses = []
        for i, b in enumerate(batches):
            if mode == 'plain':
                out = tr.step(b)
            elif mode == 'pipelined':
                out = tr.step(b, next_data=batches[i + 1]
                              if i + 1 < len(batches) else None)
            else:  # announces a batch that never comes (fresh tensors)
                out = tr.step(b, next_data=batch(100 + i))
            losses.append(out['loss'].clone())
        torch.cuda.synchronize()
        results.append((torch.stack(losses), tr.arena.flat_param.clone()))
        hits[mode] = getattr(det, 'prefetch_hits', 0)
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0])
        assert torch.equal(results[0][1], other[1])
    # with distinct batch tensors every step after the first
    # must CONSUME the prefetch (round 2's queue never hit and ran the teacher
    # twice per step); announcements that never arrive are never consumed
    assert hits['pipelined'] == len(batches) - 1, hits
    assert hits['stale'] == 0 and hits['plain'] == 0, hits


def _batch_hw(seed, num_gt, dev, hw):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(2, (hw[0] - 8, hw[1] - 10), hw, num_gt, seed)
    return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                gt_labels=[x.to(dev) for x in b['gt_labels']])


def test_auto_stepper_graph_mode_equals_eager():
    """train.AutoStepper(mode='graph'): the per-shape hipGraph path (the bf16
    default of rounds 3-4; since round 5 the eager step with the teacher's launch
    lists is GPU-bound and faster, so 'eager' is the default in every precision and
    the graph path is on request); a sequence that alternates between two
    padded shapes (the GroupSampler's two aspect groups) with different GT
    counts and a moving lr gives the SAME parameters as plain eager steps, bit
    for bit -- the capture's warm-up steps leave no trace (state saved /
    restored around them), so every batch is exactly one update."""
    from ld_amd import layers as Y
    from ld_amd.train import AutoStepper
    dev = torch.device('cuda:0')
    Y.set_precision('bf16')
    try:
        wide, tall = (128, 160), (160, 128)
        seq = [_batch_hw(41, [3, 2], dev, wide), _batch_hw(42, [5, 1], dev, tall),
               _batch_hw(43, [2, 6], dev, wide), _batch_hw(44, [1, 1], dev, wide),
               _batch_hw(45, [4, 3], dev, tall)]
        lrs = [0.01, 0.01, 0.006, 0.02, 0.01]
        eager = _trainer(dev)
        outs_e = []
        for d, lr in zip(seq, lrs):
            eager.lr = lr
            outs_e.append(float(eager.step(d)['loss']))
        torch.cuda.synchronize()
        tr = _trainer(dev)
        assert AutoStepper(tr).mode == 'eager'  # the default, bf16 included
        st = AutoStepper(tr, mode='graph', max_gt=16)
        assert st.mode == 'graph'
        outs_g = []
        for k, (d, lr) in enumerate(zip(seq, lrs)):
            tr.lr = lr
            nxt = seq[k + 1] if k + 1 < len(seq) else None
            outs_g.append(st.step(d, next_data=nxt)['loss'].clone())
        torch.cuda.synchronize()
        assert st.captures == 2  # one graph per padded shape
        assert tr.iter == eager.iter == len(seq)
        assert [float(l) for l in outs_g] == outs_e
        assert torch.equal(tr.arena.flat_param, eager.arena.flat_param)
        assert torch.equal(tr.flat_momentum, eager.flat_momentum)
    finally:
        Y.set_precision('fp32')
    # fp32: eager with the teacher one step ahead is the default
    assert AutoStepper(_trainer(dev)).mode == 'eager'


@pytest.mark.parametrize('launcher', ['graph', 'list'])
def test_auto_stepper_pipelined_equals_eager(launcher):
    from ld_amd import layers as Y
    from ld_amd.train import AutoStepper
    dev = torch.device('cuda:0')
    Y.set_precision('bf16')
    try:
        seq = [_batch_g(51, [3, 2], dev), _batch_g(52, [5, 1], dev),
               _batch_g(53, [2, 7], dev), _batch_g(54, [4, 4], dev)]
        eager = _trainer(dev)
        outs_e = [float(eager.step(d)['loss']) for d in seq]
        torch.cuda.synchronize()
        tr = _trainer(dev)
        st = AutoStepper(tr, mode='pipelined', max_gt=16, launcher=launcher)
        outs = []
        for k, d in enumerate(seq):
            nxt = seq[min(k + 1, len(seq) - 1)]
            outs.append(st.step(d, next_data=nxt)['loss'].clone())
        torch.cuda.synchronize()
        assert [float(l) for l in outs] == outs_e
        assert torch.equal(tr.arena.flat_param, eager.arena.flat_param)
        with pytest.raises(ValueError):
            st.step(seq[0], next_data=seq[1])  # not the batch the pipeline holds
    finally:
        Y.set_precision('fp32')


def test_step_list_replays_a_two_stream_capture():
    """ld_step_list_* on a small hand-made capture (not the train step): a fork onto a
    side stream, work on both, a join, a memset and a kernel-copy -- the list keeps
    the two lanes, orders them with events at the fork / join, and a replay on NEW
    input data gives what eager execution gives; a captured torch copy (a 1-D memcpy
    node this runtime cannot describe) is refused instead of replayed wrongly."""
    import ctypes as C
    from ld_amd import layers as Y
    from ld_amd import lib as L
    from ld_amd.train import _StepList
    dev = torch.device('cuda:0')
    x = torch.randn(1 << 20, device=dev)
    out = torch.empty_like(x)
    tmp = torch.empty_like(x)
    side = torch.cuda.Stream(device=dev)
    cap = torch.cuda.Stream(device=dev)

    def work():
        main = torch.cuda.current_stream(dev)
        a = x * 2.0
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b = a + 1.0
            b = b * b
        c = a * 3.0
        tmp.zero_()                      # a memset node
        Y.copy_into(tmp, c)              # ld_copy_d2d: a kernel node
        main.wait_stream(side)
        torch.add(b, tmp, out=out)

    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(cap):
        work()                           # warm-up: allocator, lazy init
    torch.cuda.current_stream(dev).wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=cap):
        work()
    sl = _StepList(g)
    assert sl.info['lanes'] == 2 and sl.info['cross_lane_waits'] >= 2, sl.info
    assert sl.info['memcpys'] == 0 and sl.info['kernels'] >= 6, sl.info
    for seed in (1, 2, 3):
        x.copy_(torch.randn(1 << 20, generator=torch.Generator().manual_seed(seed)).to(dev))
        out.fill_(float('nan'))
        sl.replay(dev)
        torch.cuda.synchronize()
        a = x * 2.0
        want = (a + 1.0) * (a + 1.0) + a * 3.0
        assert torch.equal(out, want), seed
    # a torch copy inside a capture is a hipMemcpyAsync: refused by the builder
    g2 = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g2, stream=cap):
        tmp.copy_(x)
        torch.add(tmp, 1.0, out=out)
    rc = L.get_lib().ld_step_list_build(C.c_void_p(g2.raw_cuda_graph()), 4)
    assert rc == -3, rc  # LD_EUNSUPPORTED
