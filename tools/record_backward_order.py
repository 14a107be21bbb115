"""Record the order in which the parameter gradients of the real LD train step
(GFocal-R50 <- R101) become complete during ONE backward pass on the GPU: the
sequence of GradArena._on_grad calls by parameter name.  The CPU test
tests/test_ddp_real_arena.py replays this order on 2 gloo ranks over the real
175-parameter arena (bucket completion order, one all-reduce per bucket, the
"no gradient this step" path).

    python tools/record_backward_order.py tests/golden/backward_order.json
"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out = sys.argv[1]
    from ld_amd import layers as Y
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import SGDTrainer
    dev = torch.device('cuda:0')
    res = {}
    for mode in ('fp32', 'bf16'):
        Y.set_precision(mode)
        det = model_zoo.build_seeded_ld_detector(50, 101, dev)
        tr = SGDTrainer(det, lr=0.0)
        names = {id(p): n for n, p in det.named_parameters()}
        b = synthetic.synthetic_batch(2, (256, 320), (256, 320), [5, 3], 77)
        d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
                 gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
                 gt_labels=[x.to(dev) for x in b['gt_labels']])
        tr.step(d)
        seq = []
        inner = tr.arena._on_grad

        def spy(p, inner=inner, seq=seq, arena=tr.arena):
            if id(p) not in arena._seen:
                seq.append(names[id(p)])
            return inner(p)
        for p in tr.arena.params:
            p._ld_ready = spy
        for h in tr.arena._hooks:  # autograd's post-accumulate path too
            h.remove()
        tr.arena._hooks = [p.register_post_accumulate_grad_hook(spy)
                           for p in tr.arena.params]
        tr.step(d)
        torch.cuda.synchronize()
        assert len(seq) == len(tr.arena.params), (len(seq), len(tr.arena.params))
        res[mode] = seq
    Y.set_precision('fp32')
    same = res['fp32'] == res['bf16']
    json.dump(dict(comment='order of GradArena._on_grad calls (first call per '
                           'parameter) in one backward pass of the LD step, '
                           'GFocal-R50 <- R101; tools/record_backward_order.py',
                   same_in_bf16=same, order=res['fp32'],
                   order_bf16=None if same else res['bf16']),
              open(out, 'w'), indent=0)
    print('recorded', len(res['fp32']), 'parameters; bf16 order identical:', same)
    print('first', res['fp32'][:4], 'last', res['fp32'][-4:])


if __name__ == '__main__':
    main()
