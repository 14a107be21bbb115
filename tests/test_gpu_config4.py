"""GPU test (-m gpu) of BASELINE config 4 at ITS OWN shape
(configs/ld/ld_r101_gflv1_r101dcn_fpn_coco_2x.py: ResNet-101 student <-
R101-DCN(c3-c5) teacher, 2 x 800x1344, fp32): one whole train step, and -- the
pin available without mmcv's compiled op -- with the DCN offsets at their
initial value 0 the loss table equals the plain-R101-teacher step on the same
weights (a zero-offset DCNv1 IS the convolution).  DCN with non-zero offsets:
parity unpinned (oracle/dcn_oracle.py is the only checker,
tests/test_gpu_v2.py::test_dcn_forward_vs_oracle)."""
import numpy as np
import pytest
import torch

from ld_amd import synthetic

pytestmark = pytest.mark.gpu

LOSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im']


def test_config4_full_size_step():
    from ld_amd import build_detector, model_zoo
    from ld_amd.train import SGDTrainer
    assert torch.cuda.is_available(), 'this test needs the MI355X'
    dev = torch.device('cuda:0')
    cfg = model_zoo.ld_r101_dcn_detector()
    det = build_detector(cfg)
    plain_cfg = model_zoo.ld_detector(101, 101, loss_im_weight=0.0,
                                      with_vlr_kd=False)
    plain = build_detector(plain_cfg)
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(plain.teacher_model.state_dict(), seed=2)
    det.load_state_dict(ssd)
    plain.load_state_dict(ssd)
    plain.teacher_model.load_state_dict(tsd)
    missing = det.teacher_model.load_state_dict(tsd, strict=False)
    assert all('conv_offset' in k for k in missing.missing_keys)
    assert len(missing.missing_keys) == 2 * (4 + 23 + 3)
    for k, v in det.teacher_model.state_dict().items():
        if 'conv_offset' in k:
            v.zero_()
    det.to(dev).train()
    plain.to(dev).train()
    b = synthetic.synthetic_batch(2, (800, 1333), (800, 1344), 7, 1234)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    t1 = torch.stack([torch.stack(v) for v in det(**d).values()])
    t0 = torch.stack([torch.stack(v) for v in plain(**d).values()])
    del plain
    np.testing.assert_allclose(t1.detach().cpu().numpy(),
                               t0.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    # the r18-style config names loss_ld only; LDHead's constructor defaults
    # (ld_head.py:47-63) still switch VLR-LD and KD on, the imitation term off
    tab = t1.detach().cpu().numpy()
    for k in ('loss_ld', 'loss_ld_vlr', 'loss_kd'):
        assert tab[LOSS_KEYS.index(k)].sum() > 0, k
    assert tab[LOSS_KEYS.index('loss_im')].sum() == 0
    # non-zero offsets move the distillation term only
    for k, v in det.teacher_model.state_dict().items():
        if k.endswith('conv_offset.bias'):
            v.fill_(0.6)
    t2 = torch.stack([torch.stack(v) for v in det(**d).values()])
    diff = (t2 - t1).abs().sum(1).detach().cpu().numpy()
    assert diff[LOSS_KEYS.index('loss_ld')] > 1e-4
    assert diff[LOSS_KEYS.index('loss_cls')] == 0.0
    # two optimizer steps of the whole R101 <- R101-DCN train engine
    tr = SGDTrainer(det, lr=0.0025)
    l0 = float(tr.step(d)['loss'])
    l1 = float(tr.step(d)['loss'])
    torch.cuda.synchronize()
    assert np.isfinite(l0) and np.isfinite(l1)
