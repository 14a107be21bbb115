"""TEST INFRASTRUCTURE ONLY -- plain PyTorch (CPU, fp32) restatement of the
networks on the LD path, driven by an mmdet-keyed state_dict.  It is the
"plain PyTorch fp32 reference" the HIP conv/norm stack is checked against and
the `cpu_baseline` ("port") leg of bench.py.  Never imported by `ld_amd`.

Parity status: pinned through tests/golden/e2e.npz (loss tables, feature and
gradient fingerprints produced by the reference itself with the same seeded
state_dicts) -- tests/test_oracle_golden.py::test_net_oracle_vs_golden.

Follows (reference file:line):
  ResNet            mmdet/models/backbones/resnet.py:13-299,558-637
  ResLayer          mmdet/models/utils/res_layer.py:24-102
  FPN               mmdet/models/necks/fpn.py:170-221
  GFLHead.forward   mmdet/models/dense_heads/gfl_head.py:145-183
  forward_train     mmdet/models/detectors/kd_one_stage.py:46-81
  GFocalHead.forward_single (GFLv2 quality branch)
                    mmdet/models/dense_heads/gfocal_head.py:201-217
                    (pinned through tests/golden/lossblock_v2.npz / e2e_v2.npz)
"""
import numpy as np
import torch
import torch.nn.functional as F

import ld_oracle as O

STAGE_BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3),
                101: (3, 4, 23, 3)}


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, eps)


def resnet_forward(sd, x, depth, prefix='backbone.'):
    """-> (C2, C3, C4, C5); BN always with running stats (norm_eval)."""
    bottleneck = depth >= 50
    x = F.relu(_bn(F.conv2d(x, sd[prefix + 'conv1.weight'], stride=2,
                            padding=3), sd, prefix + 'bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nblocks in enumerate(STAGE_BLOCKS[depth]):
        for b in range(nblocks):
            p = f'{prefix}layer{li + 1}.{b}.'
            stride = 2 if (b == 0 and li > 0) else 1
            identity = x
            if bottleneck:
                out = F.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight']), sd,
                                 p + 'bn1'))
                out = F.relu(_bn(F.conv2d(out, sd[p + 'conv2.weight'],
                                          stride=stride, padding=1), sd,
                                 p + 'bn2'))
                out = _bn(F.conv2d(out, sd[p + 'conv3.weight']), sd,
                          p + 'bn3')
            else:
                out = F.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight'],
                                          stride=stride, padding=1), sd,
                                 p + 'bn1'))
                out = _bn(F.conv2d(out, sd[p + 'conv2.weight'], padding=1),
                          sd, p + 'bn2')
            if p + 'downsample.0.weight' in sd:
                identity = _bn(F.conv2d(x, sd[p + 'downsample.0.weight'],
                                        stride=stride), sd,
                               p + 'downsample.1')
            x = F.relu(out + identity)
        outs.append(x)
    return tuple(outs)


def fpn_forward(sd, feats, prefix='neck.', start_level=1, num_outs=5):
    ins = feats[start_level:]
    lats = [F.conv2d(f, sd[f'{prefix}lateral_convs.{i}.conv.weight'],
                     sd[f'{prefix}lateral_convs.{i}.conv.bias'])
            for i, f in enumerate(ins)]
    for i in range(len(lats) - 1, 0, -1):
        lats[i - 1] = lats[i - 1] + F.interpolate(
            lats[i], size=lats[i - 1].shape[2:], mode='nearest')
    outs = [F.conv2d(l, sd[f'{prefix}fpn_convs.{i}.conv.weight'],
                     sd[f'{prefix}fpn_convs.{i}.conv.bias'], padding=1)
            for i, l in enumerate(lats)]
    for i in range(len(lats), num_outs):  # extra convs 'on_output'
        outs.append(F.conv2d(outs[-1], sd[f'{prefix}fpn_convs.{i}.conv.weight'],
                             sd[f'{prefix}fpn_convs.{i}.conv.bias'], stride=2,
                             padding=1))
    return tuple(outs)


def gfl_head_forward(sd, feats, prefix='bbox_head.', stacked=4, groups=32):
    cls_scores, bbox_preds = [], []
    for l, x in enumerate(feats):
        c = r = x
        for i in range(stacked):
            c = F.relu(F.group_norm(
                F.conv2d(c, sd[f'{prefix}cls_convs.{i}.conv.weight'],
                         padding=1), groups,
                sd[f'{prefix}cls_convs.{i}.gn.weight'],
                sd[f'{prefix}cls_convs.{i}.gn.bias'], 1e-5))
            r = F.relu(F.group_norm(
                F.conv2d(r, sd[f'{prefix}reg_convs.{i}.conv.weight'],
                         padding=1), groups,
                sd[f'{prefix}reg_convs.{i}.gn.weight'],
                sd[f'{prefix}reg_convs.{i}.gn.bias'], 1e-5))
        cls_scores.append(F.conv2d(c, sd[prefix + 'gfl_cls.weight'],
                                   sd[prefix + 'gfl_cls.bias'], padding=1))
        bbox_preds.append(F.conv2d(r, sd[prefix + 'gfl_reg.weight'],
                                   sd[prefix + 'gfl_reg.bias'], padding=1) *
                          sd[f'{prefix}scales.{l}.scale'])
    return cls_scores, bbox_preds


def detector_forward(sd, img, depth):
    feats = fpn_forward(sd, resnet_forward(sd, img, depth))
    cls, reg = gfl_head_forward(sd, feats)
    return feats, cls, reg


def trainable_keys(sd, frozen_stages=1):
    """Parameters that receive gradients (resnet.py:572-588: stem + layer1
    frozen; buffers excluded)."""
    out = []
    for k in sd:
        if k.endswith(('running_mean', 'running_var', 'num_batches_tracked',
                       'integral.project')):
            continue
        if k.startswith(('backbone.conv1.', 'backbone.bn1.')):
            continue
        if any(k.startswith(f'backbone.layer{i}.')
               for i in range(1, frozen_stages + 1)):
            continue
        out.append(k)
    return out


def ld_train_step(student_sd, teacher_sd, batch, student_depth, teacher_depth,
                  hp=None, with_backward=True, timings=None):
    """One LD forward (+backward) on the CPU: torch autograd for the nets,
    the numpy oracle for targets + loss block.  Returns dict(losses (8,5),
    grads {key: tensor}, feats, cls, reg).  ``timings`` (dict) receives the
    wall seconds of the stages: student_net, teacher_net, loss_block,
    backward."""
    import time as _time
    tick = [_time.perf_counter()]

    def lap(name):
        now = _time.perf_counter()
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + now - tick[0]
        tick[0] = now

    keys = trainable_keys(student_sd)
    sd = dict(student_sd)
    for k in keys:
        sd[k] = student_sd[k].detach().clone().requires_grad_(with_backward)
    img = batch['img']
    tick[0] = _time.perf_counter()
    feats, cls, reg = detector_forward(sd, img, student_depth)
    lap('student_net')
    with torch.no_grad():
        t_feats, t_cls, t_reg = detector_forward(teacher_sd, img,
                                                 teacher_depth)
    lap('teacher_net')
    sizes = [tuple(f.shape[2:]) for f in cls]
    targets = O.get_targets(sizes, batch['img_metas'],
                            [b.numpy() for b in batch['gt_bboxes']],
                            [l.numpy() for l in batch['gt_labels']])
    npy = lambda ts: [t.detach().numpy() for t in ts]  # noqa: E731
    out = O.ld_loss_block(npy(cls), npy(reg), npy(t_cls), npy(t_reg),
                          npy(feats), npy(t_feats), targets, hp,
                          with_grad=with_backward)
    lap('loss_block')
    res = dict(losses=out['losses'], feats=feats, cls=cls, reg=reg,
               targets=targets)
    if with_backward:
        heads = list(cls) + list(reg) + list(feats)
        gs = [torch.from_numpy(g) for g in out['grads']['cls'] +
              out['grads']['reg'] + out['grads']['x']]
        torch.autograd.backward(heads, gs)
        res['grads'] = {k: sd[k].grad for k in keys}
        lap('backward')
    return res


# --------------------------------------------------------------------------
# GFLv2 / LDv2
# --------------------------------------------------------------------------
def quality_tail(sd, cls_feat, bbox_pred, prefix='bbox_head.', reg_max=16,
                 topk=4):
    """gfocal_head.py:201-217: softmax over the 17 bins of each side, top-4 +
    their mean -> 20 statistics -> reg_conf (1x1 20->64, ReLU, 1x1 64->1,
    sigmoid); cls_score = sigmoid(cls_feat) * quality."""
    N, _, H, W = bbox_pred.shape
    prob = F.softmax(bbox_pred.reshape(N, 4, reg_max + 1, H, W), dim=2)
    top, _ = prob.topk(topk, dim=2)
    stat = torch.cat([top, top.mean(dim=2, keepdim=True)], dim=2)
    h = F.relu(F.conv2d(stat.reshape(N, -1, H, W),
                        sd[prefix + 'reg_conf.0.weight'],
                        sd[prefix + 'reg_conf.0.bias']))
    q = torch.sigmoid(F.conv2d(h, sd[prefix + 'reg_conf.2.weight'],
                               sd[prefix + 'reg_conf.2.bias']))
    return cls_feat.sigmoid() * q, q


def gfocal_head_forward(sd, feats, prefix='bbox_head.'):
    cls_feats, bbox_preds = gfl_head_forward(sd, feats, prefix)
    scores = [quality_tail(sd, c, r, prefix)[0]
              for c, r in zip(cls_feats, bbox_preds)]
    return scores, bbox_preds, cls_feats


def ldv2_loss_step(head_sd, hi, batch, hp=None, prefix=''):
    """LDv2Head.loss on given tower outputs: ``hi`` = dict of lists cls
    (cls_feat, 81 ch), reg, x, t_cls (teacher cls_feat), t_reg, t_x.  torch
    autograd for the quality branch, the numpy oracle for targets + loss
    block.  Returns losses (8, L), grads wrt cls_feat / reg / x and the
    reg_conf parameters."""
    names = [k for k in head_sd if k.startswith(prefix + 'reg_conf')]
    sd = dict(head_sd)
    for k in names:
        sd[k] = head_sd[k].detach().clone().requires_grad_(True)
    cf = [t.detach().clone().requires_grad_(True) for t in hi['cls']]
    rg = [t.detach().clone().requires_grad_(True) for t in hi['reg']]
    tails = [quality_tail(sd, c, r, prefix) for c, r in zip(cf, rg)]
    scores = [t[0] for t in tails]
    sizes = [tuple(f.shape[2:]) for f in cf]
    targets = O.get_targets(sizes, batch['img_metas'],
                            [b.numpy() for b in batch['gt_bboxes']],
                            [l.numpy() for l in batch['gt_labels']])
    npy = lambda ts: [t.detach().numpy() for t in ts]  # noqa: E731
    out = O.ld_loss_block(npy(scores), npy(rg), None, npy(hi['t_reg']),
                          npy(hi['x']), npy(hi['t_x']), targets, hp,
                          kd=(npy(cf), npy(hi['t_cls'])))
    g = out['grads']
    # cls_feat gets the KD gradient directly and the QFL one through the tail;
    # reg gets the loss gradient directly and the quality one through the tail
    torch.autograd.backward(scores, [torch.from_numpy(a) for a in g['cls']])
    gcf = [c.grad + torch.from_numpy(a) for c, a in zip(cf, g['kd'])]
    grg = [r.grad + torch.from_numpy(a) for r, a in zip(rg, g['reg'])]
    return dict(losses=out['losses'], g_cls_feat=gcf, g_reg=grg,
                g_x=[torch.from_numpy(a) for a in g['x']],
                g_params={k: sd[k].grad for k in names},
                quality=[t[1].detach() for t in tails], scores=scores)


def ldv2_train_step(student_sd, teacher_sd, batch, student_depth,
                    teacher_depth, hp=None, with_backward=True):
    """One LDv2 forward (+backward) on the CPU (kd_one_stage.py:46-81 with an
    LDv2Head student and a GFocalHead teacher)."""
    keys = trainable_keys(student_sd)
    sd = dict(student_sd)
    for k in keys:
        sd[k] = student_sd[k].detach().clone().requires_grad_(with_backward)
    img = batch['img']
    feats = fpn_forward(sd, resnet_forward(sd, img, student_depth))
    scores, reg, cls_feat = gfocal_head_forward(sd, feats)
    with torch.no_grad():
        t_feats = fpn_forward(teacher_sd,
                              resnet_forward(teacher_sd, img, teacher_depth))
        _, t_reg, t_cls_feat = gfocal_head_forward(teacher_sd, t_feats)
    sizes = [tuple(f.shape[2:]) for f in scores]
    targets = O.get_targets(sizes, batch['img_metas'],
                            [b.numpy() for b in batch['gt_bboxes']],
                            [l.numpy() for l in batch['gt_labels']])
    npy = lambda ts: [t.detach().numpy() for t in ts]  # noqa: E731
    out = O.ld_loss_block(npy(scores), npy(reg), None, npy(t_reg), npy(feats),
                          npy(t_feats), targets, hp, with_grad=with_backward,
                          kd=(npy(cls_feat), npy(t_cls_feat)))
    res = dict(losses=out['losses'], feats=feats, scores=scores, reg=reg,
               cls_feat=cls_feat, targets=targets)
    if with_backward:
        g = out['grads']
        heads = list(scores) + list(reg) + list(feats) + list(cls_feat)
        gs = [torch.from_numpy(a) for a in g['cls'] + g['reg'] + g['x'] +
              g['kd']]
        torch.autograd.backward(heads, gs)
        res['grads'] = {k: sd[k].grad for k in keys}
    return res
