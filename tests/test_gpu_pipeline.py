"""GPU tests (-m gpu) of the device input pipeline (csrc/pipeline.hip through
ld_preprocess_batch) against oracle/pipeline_oracle.py: bit-exact -- the
resize is integer arithmetic and the normalisation is one fp32 subtract and
one fp32 multiply, compiled with -ffp-contract=off."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))

pytestmark = pytest.mark.gpu

MEAN, STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)


def _images(rs, shapes):
    out = []
    for h, w in shapes:
        # smooth + noise so that interpolation errors are visible but realistic
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.sin(yy / 7.0)[..., None] * 60 + np.cos(xx / 5.0)[..., None] *
                60 + 128 + rs.randn(h, w, 3) * 25)
        out.append(np.clip(base, 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize('shapes,scale', [
    ([(480, 640), (427, 640)], (1333, 800)),       # the C2 batch geometry
    ([(640, 480), (375, 500), (333, 1000)], (1333, 800)),
    ([(37, 53)], (53, 37)),                          # identity size
    ([(64, 96), (96, 64)], (48, 32)),                # 2x downsample
    ([(5, 7), (1, 1), (2, 300)], (1333, 800)),       # degenerate sources
])
def test_preprocess_bit_exact(shapes, scale):
    import pipeline_oracle as PO
    from ld_amd.pipeline import DevicePipeline
    rs = np.random.RandomState(len(shapes) * 13 + scale[0])
    imgs = _images(rs, shapes)
    boxes = [np.array([[1, 2, 30, 20], [0, 0, w, h]], np.float32)
             for h, w in shapes]
    labels = [np.array([3, 7]) for _ in shapes]
    pipe = DevicePipeline(img_scale=scale, mean=MEAN, std=STD, to_rgb=True,
                          size_divisor=32, device='cuda:0')
    np.random.seed(3)
    plans = pipe.plan([im.shape[:2] for im in imgs])
    plans[0]['flip'], plans[-1]['flip'] = True, len(shapes) == 1
    out = pipe(imgs, boxes, labels, plans=plans)
    torch.cuda.synchronize()
    got = out['img'].cpu().numpy()
    N, _, Hp, Wp = got.shape
    assert N == len(shapes) and Hp % 32 == 0 and Wp % 32 == 0
    for i, (im, p) in enumerate(zip(imgs, plans)):
        nh, nw = p['img_shape'][:2]
        ref = PO.preprocess(im, nh, nw, p['flip'], MEAN, STD, True, Hp, Wp)
        np.testing.assert_array_equal(got[i], ref, err_msg=f'image {i}')
        m = out['img_metas'][i]
        assert m['img_shape'] == (nh, nw, 3) and m['flip'] == p['flip']
        assert m['pad_shape'][0] % 32 == 0 and m['pad_shape'][0] >= nh
        b = out['gt_bboxes'][i].cpu().numpy()
        assert b.dtype == np.float32 and (b[:, 2] <= nw).all()
        assert out['gt_labels'][i].dtype == torch.int64


def test_preprocess_to_rgb_false_and_unit_norm():
    """to_rgb=False keeps BGR order; mean 0 / std 1 returns the resized uint8
    image itself (the integer stage on its own)."""
    import pipeline_oracle as PO
    from ld_amd.pipeline import DevicePipeline
    rs = np.random.RandomState(4)
    im = _images(rs, [(120, 200)])[0]
    pipe = DevicePipeline(img_scale=(333, 200), mean=(0, 0, 0), std=(1, 1, 1),
                          to_rgb=False, flip_ratio=0.0, device='cuda:0')
    out = pipe([im])
    nh, nw = out['img_metas'][0]['img_shape'][:2]
    got = out['img'][0, :, :nh, :nw].cpu().numpy()
    ref = PO.resize_linear_u8(im, nh, nw).transpose(2, 0, 1).astype(np.float32)
    np.testing.assert_array_equal(got, ref)
    assert out['img_metas'][0]['flip'] is False


def test_preprocess_feeds_the_detector():
    """The batch the device pipeline produces is what forward_train takes: one
    R18 <- R18 LD step on it runs and gives finite losses."""
    from ld_amd import model_zoo
    from ld_amd.pipeline import DevicePipeline
    rs = np.random.RandomState(6)
    imgs = _images(rs, [(120, 160), (107, 160)])
    boxes = [np.array([[10, 12, 90, 100], [50, 20, 150, 110]], np.float32),
             np.array([[5, 5, 80, 60]], np.float32)]
    labels = [np.array([1, 17]), np.array([33])]
    pipe = DevicePipeline(img_scale=(320, 192), device='cuda:0')
    np.random.seed(1)
    batch = pipe(imgs, boxes, labels)
    det = model_zoo.build_seeded_ld_detector(18, 18, torch.device('cuda:0'))
    losses = det(img=batch['img'], img_metas=batch['img_metas'],
                 gt_bboxes=batch['gt_bboxes'], gt_labels=batch['gt_labels'])
    loss, _ = det._parse_losses(losses)
    loss.backward()
    assert np.isfinite(float(loss.detach()))
