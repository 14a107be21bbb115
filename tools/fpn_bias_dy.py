"""VERDICT r4 next #6 (GPU side): the gradient arriving at the FPN outputs P5 / P6
(= dy of neck.fpn_convs.2 / .3, whose bias gradients are its plain sums) on the HIP
path against the oracle nets in float64 and float32 (oracle/_ref/fpn_dy64_c2.npz,
tools/gen_fpn_dy64.py).  Prints, per level: the bias-gradient error of the HIP path
and of the fp32 oracle vs float64, and how the element-wise dy error is distributed
(a handful of large elements = threshold flips upstream; a uniform floor = summation
noise).
    python tools/fpn_bias_dy.py > profiles/r05_fpn_bias_dy.txt     (through gpurun)"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ld_amd import build_detector, model_zoo, synthetic  # noqa: E402

path = os.path.join(REPO, 'oracle', '_ref', 'fpn_dy64_c2.npz')
if not os.path.exists(path):
    sys.exit('oracle/_ref/fpn_dy64_c2.npz missing: run tools/gen_fpn_dy64.py in the '
             'build container first')
ref = np.load(path)
dev = torch.device('cuda:0')
ge = np.load(os.path.join(REPO, 'tests', 'golden', 'e2e.npz'))
cfg = ge['c2_r50_cfg']
pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
num_gt = [int(x) for x in ge['c2_r50_num_gt']]
batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
det = build_detector(model_zoo.ld_detector(50, 101))
det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(), seed=1))
det.teacher_model.load_state_dict(
    synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2))
det = det.to(dev)
det.train()
got = {}
neck_forward = det.neck.forward


def hooked(inputs):
    outs = neck_forward(inputs)
    for lvl in (2, 3):
        outs[lvl].register_hook(lambda g, lvl=lvl: got.__setitem__(lvl, g.detach().clone()))
    return outs


det.neck.forward = hooked
losses = det(img=batch['img'].to(dev), img_metas=batch['img_metas'],
             gt_bboxes=[b.to(dev) for b in batch['gt_bboxes']],
             gt_labels=[l.to(dev) for l in batch['gt_labels']])
loss, _ = det._parse_losses(losses)
loss.backward()
torch.cuda.synchronize()
params = dict(det.named_parameters())
for lvl in (2, 3):
    d64 = ref[f'dy_f64_l{lvl}']
    d32 = ref[f'dy_f32_l{lvl}'].astype(np.float64)
    ours = got[lvl].double().cpu().numpy().reshape(d64.shape)
    scale = np.abs(d64).max()
    b64 = d64.sum((0, 2, 3))
    bscale = np.abs(b64).max()
    name = f'neck.fpn_convs.{lvl}.conv.bias'
    ours_b = params[name].grad.double().cpu().numpy()
    print(f'== level {lvl} ({name}), dy shape {d64.shape}, max|dy| {scale:.3e}, '
          f'max|bias grad| {bscale:.3e}')
    for tag, arr in (('HIP path', ours), ('fp32 oracle', d32)):
        e = np.abs(arr - d64)
        be = np.abs(arr.sum((0, 2, 3)) - b64)
        big = e > 50 * np.median(e[e > 0]) if (e > 0).any() else e > 0
        order = np.argsort(e.reshape(-1))[::-1][:5]
        top = [(tuple(int(v) for v in np.unravel_index(i, e.shape)),
                float(e.reshape(-1)[i] / scale)) for i in order]
        print(f'  {tag:11s}: bias-grad err / max|bias grad| {be.max() / bscale:.2e} '
              f'(channel {int(be.argmax())}); dy err / max|dy|: median '
              f'{np.median(e) / scale:.2e}, max {e.max() / scale:.2e}, elements > 50 x '
              f'median: {int(big.sum())} of {e.size}')
        print(f'               largest dy errors (n, c, h, w), err / max|dy|: {top}')
        # how much of the worst channel's bias error the few large elements explain
        c = int(be.argmax())
        ec = (arr - d64)[:, c]
        idx = np.argsort(np.abs(ec).reshape(-1))[::-1][:8]
        print(f'               channel {c}: bias error {ec.sum():+.3e}; its 8 largest dy '
              f'errors sum to {ec.reshape(-1)[idx].sum():+.3e}')
    print(f'  the HIP bias gradient itself vs float64: '
          f'{np.abs(ours_b - b64).max() / bscale:.2e} of max')
