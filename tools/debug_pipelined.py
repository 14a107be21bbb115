"""Bisect helper: PipelinedGraphedStep vs eager in bf16 under feature toggles."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def batch(seed, num_gt, dev):
    from ld_amd import synthetic
    b = synthetic.synthetic_batch(2, (128, 150), (128, 160), num_gt, seed)
    return dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                gt_labels=[x.to(dev) for x in b['gt_labels']])


def trainer(dev):
    from ld_amd import model_zoo
    from ld_amd.train import SGDTrainer
    det = model_zoo.build_seeded_ld_detector(18, 18, dev, loss_im_weight=2.0)
    return SGDTrainer(det, lr=0.01)


def main():
    from ld_amd import layers as Y
    from ld_amd.train import GraphedStep, PipelinedGraphedStep
    dev = torch.device('cuda:0')
    mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    Y.set_precision(mode)
    b = [batch(31, [3, 2], dev), batch(32, [5, 1], dev), batch(33, [2, 7], dev),
         batch(34, [4, 4], dev), batch(35, [1, 1], dev)]
    seq = [b[0], b[1], b[0], b[1], b[2], b[3]]
    eager = trainer(dev)
    snaps = []
    for d in seq:
        eager.step(d)
        snaps.append(eager.arena.flat_param.clone())
    torch.cuda.synchronize()
    # eager twice: is the eager path itself reproducible?
    e2 = trainer(dev)
    for d in seq:
        e2.step(d)
    torch.cuda.synchronize()
    print('eager vs eager max diff',
          float((e2.arena.flat_param - eager.arena.flat_param).abs().max()))
    tr = trainer(dev)
    ps = PipelinedGraphedStep(tr, b[0], b[1], warmup=1, max_gt=16)
    torch.cuda.synchronize()
    print('after warm-up (2 eager steps) diff',
          float((tr.arena.flat_param - snaps[1]).abs().max()))
    for i, nb in enumerate([b[1], b[2], b[3], b[4]]):
        ps.step(nb)
        torch.cuda.synchronize()
        d = (tr.arena.flat_param - snaps[2 + i]).abs()
        print(f'replay {i}: max diff {float(d.max()):.3e}, '
              f'{int((d > 0).sum())} of {d.numel()} differ', flush=True)
    # plain GraphedStep with the same toggles
    tr2 = trainer(dev)
    e3 = trainer(dev)
    for d in (b[0], b[0], b[0], b[2]):
        e3.step(d)
    g = GraphedStep(tr2, batch(31, [3, 2], dev), warmup=2, max_gt=16)
    g.replay()
    g.copy_inputs(b[2])
    g.replay()
    torch.cuda.synchronize()
    print('GraphedStep vs eager max diff',
          float((tr2.arena.flat_param - e3.arena.flat_param).abs().max()))


if __name__ == '__main__':
    main()
