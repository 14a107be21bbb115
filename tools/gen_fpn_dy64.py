"""why are neck.fpn_convs.{2,3}.conv.bias 100x further from
the float64 gradients than the reference is?  A conv bias gradient is the plain
sum of dy over (n, h, w), so the answer is in dy.  This script (CPU, build
container) evaluates the c2_r50 step of tests/golden/e2e.npz with the ORACLE nets
in float64 and in float32 and stores the gradient arriving at the FPN outputs P5 /
P6 (the dy of fpn_convs.2 / .3) under oracle/_ref/ (git-ignored, shipped to the GPU
box); tools/fpn_bias_dy.py compares the HIP path's dy with both there.
    python tools/gen_fpn_dy64.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import ld_oracle as O  # noqa: E402
import net_oracle as NO  # noqa: E402
from ld_amd import build_detector, model_zoo, synthetic  # noqa: E402


def step(dtype):
    ge = np.load(os.path.join(REPO, 'tests', 'golden', 'e2e.npz'))
    name = 'c2_r50'
    cfg = ge[name + '_cfg']
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
    num_gt = [int(x) for x in ge[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    det = build_detector(model_zoo.ld_detector(50, 101))
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2)
    cast = lambda sd: {k: (v.to(dtype) if v.is_floating_point() else v)  # noqa: E731
                       for k, v in sd.items()}
    ssd, tsd = cast(ssd), cast(tsd)
    keys = NO.trainable_keys(ssd)
    sd = dict(ssd)
    for k in keys:
        sd[k] = ssd[k].detach().clone().requires_grad_(True)
    img = batch['img'].to(dtype)
    feats, cls, reg = NO.detector_forward(sd, img, 50)
    for f in feats:
        f.retain_grad()
    with torch.no_grad():
        t_feats, t_cls, t_reg = NO.detector_forward(tsd, img, 101)
    sizes = [tuple(f.shape[2:]) for f in cls]
    targets = O.get_targets(sizes, batch['img_metas'],
                            [b.numpy() for b in batch['gt_bboxes']],
                            [l.numpy() for l in batch['gt_labels']])
    npy = lambda ts: [t.detach().numpy() for t in ts]  # noqa: E731
    out = O.ld_loss_block(npy(cls), npy(reg), npy(t_cls), npy(t_reg), npy(feats),
                          npy(t_feats), targets, None, with_grad=True)
    heads = list(cls) + list(reg) + list(feats)
    gs = [torch.from_numpy(g).to(dtype) for g in out['grads']['cls'] +
          out['grads']['reg'] + out['grads']['x']]
    torch.autograd.backward(heads, gs)
    return [f.grad.detach().double().numpy() for f in feats], \
        {k: sd[k].grad.detach().double().numpy() for k in keys
         if 'fpn_convs' in k and k.endswith('bias')}


def main():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    out = {}
    for tag, dt in (('f64', torch.float64), ('f32', torch.float32)):
        t0 = time.time()
        dys, biases = step(dt)
        print(tag, 'done in %.0f s' % (time.time() - t0), flush=True)
        for lvl in (2, 3):
            out[f'dy_{tag}_l{lvl}'] = dys[lvl].astype(np.float64 if tag == 'f64'
                                                       else np.float32)
        for k, v in biases.items():
            out[f'bias_{tag}_{k}'] = v
    os.makedirs(os.path.join(REPO, 'oracle', '_ref'), exist_ok=True)
    path = os.path.join(REPO, 'oracle', '_ref', 'fpn_dy64_c2.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1e6, 'MB')


if __name__ == '__main__':
    main()
