"""GPU tests (-m gpu) of the frozen teacher as recorded launch lists
(ld_record_* in include/ld_hip.h, KnowledgeDistillationSingleStageDetector.
_teacher_replay): a replayed forward must equal the ordinary forward BIT FOR BIT on
fresh images, slots must rotate without clobbering results still in use, a change
of the teacher's weights must invalidate the lists, and whole train steps with the
teacher one step ahead must leave identical parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flat(tx, out):
    return list(tx) + [t for lvl in out for t in lvl]


def _f(t):
    return t.float() if not isinstance(t, torch.Tensor) else t


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_teacher_replay_equals_plain_forward(precision, monkeypatch):
    from ld_amd import layers as Y
    from ld_amd import model_zoo
    dev = torch.device('cuda:0')
    prev = Y.get_precision()
    Y.set_precision(precision)
    try:
        det = model_zoo.build_seeded_ld_detector(50, 101, dev)
        g = torch.Generator().manual_seed(5)
        imgs = [torch.randn(2, 3, 128, 160, generator=g).to(dev) for _ in range(9)]
        monkeypatch.setenv('LD_TEACHER_REPLAY', '0')
        want = []
        for im in imgs:
            tx, out = det._teacher_forward(im)
            want.append([_f(t).clone() for t in _flat(tx, out)])
        monkeypatch.setenv('LD_TEACHER_REPLAY', '1')
        held = []
        for i, im in enumerate(imgs):
            tx, out = det._teacher_forward(im)
            got = [_f(t) for t in _flat(tx, out)]
            for a, b in zip(got, want[i]):
                assert torch.equal(a, b), (precision, i)
            held.append((i, _flat(tx, out)))
            # the result of the PREVIOUS call is still intact (slots rotate)
            if i >= 1:
                j, prev_out = held[i - 1]
                for a, b in zip(prev_out, want[j]):
                    assert torch.equal(_f(a), b), (precision, 'clobbered', i)
        # 1 warm-up + 3 recordings, then replays
        assert det.teacher_replays == len(imgs) - 1 - det.TEACHER_SLOTS
        pl, = det._tplans.values()
        assert len(pl['slots']) == det.TEACHER_SLOTS
        assert all(sl['launches'] > 50 for sl in pl['slots'])
        # a changed teacher weight invalidates the lists
        with torch.no_grad():
            det.teacher_model.bbox_head.gfl_cls.bias.add_(0.25)
        monkeypatch.setenv('LD_TEACHER_REPLAY', '0')
        tx, out = det._teacher_forward(imgs[0])
        ref = [_f(t).clone() for t in _flat(tx, out)]
        assert not torch.equal(ref[len(tx)], want[0][len(tx)])
        monkeypatch.setenv('LD_TEACHER_REPLAY', '1')
        for _ in range(6):
            tx, out = det._teacher_forward(imgs[0])
            for a, b in zip(_flat(tx, out), ref):
                assert torch.equal(_f(a), b)
        torch.cuda.synchronize()
    finally:
        Y.set_precision(prev)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_train_steps_with_teacher_replay_bit_identical(precision, monkeypatch):
    from ld_amd import layers as Y
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import SGDTrainer
    dev = torch.device('cuda:0')
    prev = Y.get_precision()
    Y.set_precision(precision)

    def batch(seed, gts):
        b = synthetic.synthetic_batch(2, (128, 160), (128, 160), gts, seed)
        return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                    gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                    gt_labels=[x.to(dev) for x in b['gt_labels']])

    seq = [batch(40 + i, [2 + i % 3, 1 + i % 2]) for i in range(9)]

    def run(flag):
        monkeypatch.setenv('LD_TEACHER_REPLAY', flag)
        det = model_zoo.build_seeded_ld_detector(18, 18, dev, loss_im_weight=2.0)
        tr = SGDTrainer(det, lr=0.01)
        losses = []
        for i, d in enumerate(seq):
            nxt = seq[i + 1] if i + 1 < len(seq) else None
            out = tr.step(d, next_data=nxt)
            losses.append(float(out['log_vars']['loss']))
        torch.cuda.synchronize()
        return tr.arena.flat_param.clone(), losses, getattr(det, 'teacher_replays', 0), \
            getattr(det, 'prefetch_hits', 0)

    try:
        p0, l0, r0, _ = run('0')
        p1, l1, r1, hits = run('1')
    finally:
        Y.set_precision(prev)
    assert r0 == 0 and r1 >= 4 and hits >= len(seq) - 2
    assert l0 == l1
    assert torch.equal(p0, p1)
