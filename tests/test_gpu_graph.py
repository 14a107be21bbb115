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


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_graphed_step_equals_eager(mode):
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
        g = GraphedStep(tr, static, warmup=2)  # two eager steps on d1
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

    batches = [batch(s) for s in (5, 6, 7)]
    results = []
    for mode in ('plain', 'pipelined', 'stale'):
        det = model_zoo.build_seeded_ld_detector(18, 18, dev)
        tr = SGDTrainer(det, lr=0.01)
        losses = []
        for i, b in enumerate(batches):
            if mode == 'plain':
                out = tr.step(b)
            elif mode == 'pipelined':
                out = tr.step(b, next_data=batches[(i + 1) % 3])
            else:  # announces a batch that never comes
                out = tr.step(b, next_data=batches[(i + 2) % 3])
            losses.append(out['loss'].clone())
        torch.cuda.synchronize()
        results.append((torch.stack(losses), tr.arena.flat_param.clone()))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0])
        assert torch.equal(results[0][1], other[1])
