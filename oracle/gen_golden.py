"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by EXECUTING THE
REFERENCE CODE at /root/reference (unmodified, imported through
oracle/ref_shim.py) on CPU in fp32.  Run from the repo root in the build
container:

    python oracle/gen_golden.py [--only kat,anchors,targets,lossblock,e2e,infer]

The fixtures it writes are committed; the GPU box has no /root/reference and
only ever reads the .npz files.  Inputs that can be regenerated from a seed
(`ld_amd.synthetic`) are NOT stored, only the seeds and the reference outputs.

Reference entry points exercised (file:line under /root/reference):
  mmdet/models/losses/kd_loss.py:10-88          KnowledgeDistillationKLDivLoss
  mmdet/models/losses/gfocal_loss.py:8-179      QFL / DFL
  mmdet/models/losses/iou_loss.py:85-102,325-360 GIoULoss
  mmdet/models/dense_heads/gfl_head.py:15-44    Integral
  mmdet/core/bbox/iou_calculators/iou2d_calculator.py:43-188 bbox_overlaps
  mmdet/core/anchor/anchor_generator.py:9-346   AnchorGenerator
  mmdet/core/bbox/assigners/atss_assigner.py:33-298 assign / get_vlr_region
  mmdet/models/dense_heads/ld_head.py:116-611   LDHead.loss / get_targets
  mmdet/models/detectors/kd_one_stage.py:46-81  forward_train
  mmdet/models/detectors/base.py:185-218        _parse_losses
  mmdet/models/dense_heads/gfl_head.py:354-451  GFLHead._get_bboxes (+ multiclass_nms)
  mmdet/models/dense_heads/gfocal_head.py:145-217 GFocalHead.forward_single (GFLv2)
  mmdet/models/dense_heads/ld_gflv2.py:116-380  LDv2Head.loss / loss_single
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_shim  # noqa: E402

ref_shim.install()

from ld_amd import synthetic  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
LOSS_KEYS = [
    'loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr', 'loss_kd',
    'loss_kd_neg', 'loss_im'
]


def _np(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- KATs ----
def gen_kat():
    from mmdet.core.bbox.iou_calculators import bbox_overlaps
    from mmdet.core.bbox.transforms import bbox2distance, distance2bbox
    from mmdet.models.dense_heads.gfl_head import Integral
    from mmdet.models.losses import (DistributionFocalLoss, GIoULoss,
                                     KnowledgeDistillationKLDivLoss,
                                     QualityFocalLoss)
    from mmdet.models.losses.kd_loss import IMLoss
    d = {}
    i = torch.arange(4, dtype=torch.float32)[:, None]
    j = torch.arange(17, dtype=torch.float32)[None, :]
    pred = (3 * torch.sin(0.37 * i + 0.11 * j)).requires_grad_(True)
    soft = 3 * torch.cos(0.23 * i + 0.19 * j)
    w = torch.tensor([1, .5, .25, 2.])
    d['kl_pred'], d['kl_soft'], d['kl_w'] = _np(pred), _np(soft), _np(w)
    # KAT1: LD KL, T=10, lw .25
    kl = KnowledgeDistillationKLDivLoss(loss_weight=0.25, T=10)
    d['kat1_none'] = _np(kl(pred, soft, reduction_override='none'))
    l = kl(pred, soft, weight=w, avg_factor=4.0)
    d['kat1_mean'] = _np(l)
    g, = torch.autograd.grad(l, pred)
    d['kat1_grad'] = _np(g)
    # KAT1b: KD T=2 lw 10
    kd = KnowledgeDistillationKLDivLoss(loss_weight=10, T=2)
    l = kd(pred, soft, weight=torch.ones(4), avg_factor=4)
    d['kat1b_mean'] = _np(l)
    d['kat1b_grad'] = _np(torch.autograd.grad(l, pred)[0])
    # KAT2: DFL
    dfl = DistributionFocalLoss(loss_weight=0.25)
    lab = torch.tensor([0, 3.25, 15.9, 7.5])
    d['dfl_label'] = _np(lab)
    d['kat2_none'] = _np(dfl(pred, lab, reduction_override='none'))
    l = dfl(pred, lab, weight=w, avg_factor=4.0)
    d['kat2_mean'] = _np(l)
    d['kat2_grad'] = _np(torch.autograd.grad(l, pred)[0])
    # KAT3: Integral
    integ = Integral(16)
    e = integ(pred.reshape(1, 68))
    d['kat3_integral'] = _np(e)
    d['kat3_grad'] = _np(
        torch.autograd.grad((e * torch.tensor([1., 2., 3., 4.])).sum(),
                            pred)[0])
    # KAT4: QFL
    cp = (2 * torch.sin(0.5 * torch.arange(6.)[:, None] +
                        0.3 * torch.arange(5.)[None, :])).requires_grad_(True)
    labels = torch.tensor([0, 5, 2, 5, 4, 5])
    score = torch.tensor([.7, 0, .3, 0, .9, 0])
    d['qfl_pred'], d['qfl_labels'], d['qfl_score'] = _np(cp), _np(
        labels), _np(score)
    qfl = QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=1.0)
    d['kat4_none'] = _np(qfl(cp, (labels, score), reduction_override='none'))
    l = qfl(cp, (labels, score), weight=torch.ones(6), avg_factor=2.5)
    d['kat4_mean'] = _np(l)
    d['kat4_grad'] = _np(torch.autograd.grad(l, cp)[0])
    # KAT5: GIoU
    b1 = torch.tensor([[10., 10, 30, 40], [5, 5, 15, 25],
                       [0, 0, 8, 8]]).requires_grad_(True)
    b2 = torch.tensor([[12., 8, 28, 36], [10, 10, 20, 20], [20, 20, 30, 30]])
    gw = torch.tensor([.5, .2, .9])
    d['giou_b1'], d['giou_b2'], d['giou_w'] = _np(b1), _np(b2), _np(gw)
    giou = GIoULoss(loss_weight=2.0)
    d['kat5_none'] = _np(giou(b1, b2, reduction_override='none'))
    l = giou(b1, b2, weight=gw, avg_factor=1.0)
    d['kat5_mean'] = _np(l)
    d['kat5_grad'] = _np(torch.autograd.grad(l, b1)[0])
    # KAT6: IoU aligned, pairwise iou / iof / giou / diou
    d['kat6_iou_aligned'] = _np(bbox_overlaps(b1, b2, is_aligned=True))
    for mode in ('iou', 'iof', 'giou', 'diou'):
        d['kat6_pair_' + mode] = _np(bbox_overlaps(b1, b2, mode=mode))
    # distance2bbox / bbox2distance
    pts = torch.tensor([[10.5, 20.25], [3., 4.]])
    dist = torch.tensor([[1.5, 2.5, 3.5, 4.5], [0.1, 20., 7.7, 0.]])
    d['d2b_points'], d['d2b_dist'] = _np(pts), _np(dist)
    bx = distance2bbox(pts, dist)
    d['d2b_out'] = _np(bx)
    d['b2d_out'] = _np(bbox2distance(pts, bx, max_dis=16))
    # IMLoss
    g = torch.Generator().manual_seed(3)
    xa = torch.randn(7, 256, generator=g).requires_grad_(True)
    xb = torch.randn(7, 256, generator=g)
    im = IMLoss(loss_weight=2.0)
    l = im(xa, xb)
    d['im_a'], d['im_b'], d['im_loss'] = _np(xa), _np(xb), _np(l)
    d['im_grad'] = _np(torch.autograd.grad(l, xa)[0])
    # weighted_loss doctest semantics (losses/utils.py:68-85)
    np.savez_compressed(os.path.join(OUT, 'kat_losses.npz'), **d)
    print('kat_losses.npz', len(d), 'arrays')


# -------------------------------------------------------------- anchors ----
def _anchor_generator():
    from mmdet.core.anchor import build_anchor_generator
    return build_anchor_generator(
        dict(
            type='AnchorGenerator',
            ratios=[1.0],
            octave_base_scale=8,
            scales_per_octave=1,
            strides=[8, 16, 32, 64, 128]))


def gen_anchors():
    ag = _anchor_generator()
    d = {}
    d['base_anchors'] = np.stack([_np(b) for b in ag.base_anchors])
    for name, sizes, pad in [
        ('kat7', [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)], (64, 64)),
        ('small', synthetic.level_shapes((160, 224)), (160, 224)),
        ('c1', synthetic.level_shapes((800, 800)), (800, 800)),
        ('c2', synthetic.level_shapes((800, 1344)), (800, 1344)),
    ]:
        anchors = ag.grid_anchors(sizes, device='cpu')
        flat = torch.cat(anchors)
        d[name + '_sizes'] = np.array(sizes)
        if flat.shape[0] < 2000:
            d[name + '_anchors'] = _np(flat)
        else:  # big grids: store strided samples + a float64 checksum
            idx = np.arange(0, flat.shape[0], 97)
            d[name + '_anchor_idx'] = idx
            d[name + '_anchor_samples'] = _np(flat)[idx]
            d[name + '_anchor_sum'] = _np(flat).astype(np.float64).sum(0)
        # valid flags for an image narrower than the pad (real-batch case)
        vf = ag.valid_flags(sizes, (pad[0] - 40, pad[1] - 70), device='cpu')
        d[name + '_valid_counts'] = np.array([int(v.sum()) for v in vf])
        d[name + '_valid_pad'] = np.array([pad[0] - 40, pad[1] - 70])
    np.savez_compressed(os.path.join(OUT, 'anchors.npz'), **d)
    print('anchors.npz')


# -------------------------------------------------------------- targets ----
def _ld_head(imitation_method='finegrained'):
    from mmdet.models import build_head
    cfg = dict(
        type='LDHead',
        num_classes=80,
        in_channels=256,
        stacked_convs=4,
        feat_channels=256,
        anchor_generator=dict(
            type='AnchorGenerator',
            ratios=[1.0],
            octave_base_scale=8,
            scales_per_octave=1,
            strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(
            type='QualityFocalLoss',
            use_sigmoid=True,
            beta=2.0,
            loss_weight=1.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
        loss_ld=dict(
            type='KnowledgeDistillationKLDivLoss', loss_weight=0.25, T=10),
        loss_ld_vlr=dict(
            type='KnowledgeDistillationKLDivLoss', loss_weight=0.25, T=10),
        loss_kd=dict(
            type='KnowledgeDistillationKLDivLoss', loss_weight=10, T=2),
        loss_im=dict(type='IMLoss', loss_weight=2.0),
        reg_max=16,
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        imitation_method=imitation_method,
        train_cfg=ref_shim.ConfigDict(
            assigner=dict(type='ATSSAssigner', topk=9),
            allowed_border=-1,
            pos_weight=-1,
            debug=False),
        test_cfg=ref_shim.ConfigDict(
            nms_pre=1000,
            min_bbox_size=0,
            score_thr=0.05,
            nms=dict(type='nms', iou_threshold=0.6),
            max_per_img=100))
    return build_head(cfg)


TARGET_CASES = [
    # name, pad_shape, img_shape, num_gt per image, seed
    ('small_g3', (160, 224), (160, 224), [3, 1], 11),
    ('small_g20', (160, 224), (150, 200), [20, 7], 12),
    ('c1_g7', (800, 800), (800, 800), [7, 7], 13),
    ('c2_g7', (800, 1344), (800, 1333), [7, 7], 1234),
    ('c2_g1_g40', (800, 1344), (800, 1333), [1, 40], 15),
    ('c2_g100', (800, 1344), (800, 1333), [100, 3], 16),
]


def _targets_reference(head, sizes, batch):
    """Run the reference get_anchors/get_targets on CPU."""
    anchor_list, valid_flag_list = head.get_anchors(
        sizes, batch['img_metas'], device='cpu')
    res = head.get_targets(
        anchor_list,
        valid_flag_list,
        batch['gt_bboxes'],
        batch['img_metas'],
        gt_bboxes_ignore_list=None,
        gt_labels_list=batch['gt_labels'],
        label_channels=head.cls_out_channels)
    return res


def gen_targets():
    head = _ld_head()
    d = {}
    # KAT7: tiny grid with grid-aligned GTs, CPU-vs-CPU only
    from mmdet.core.bbox.assigners import ATSSAssigner
    ag = _anchor_generator()
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    anchors = torch.cat(ag.grid_anchors(sizes, device='cpu'))
    gts = torch.tensor([[10., 12, 40, 44], [30, 5, 60, 30]])
    gl = torch.tensor([3, 17])
    assigner = ATSSAssigner(topk=9)
    nl = [64, 16, 4, 1, 1]
    ar = assigner.assign(anchors, nl, gts, None, gl)
    d['kat7_gt'], d['kat7_labels'] = _np(gts), _np(gl)
    d['kat7_gt_inds'] = _np(ar.gt_inds)
    d['kat7_max_overlaps'] = _np(ar.max_overlaps)
    d['kat7_vlr'] = _np(assigner.get_vlr_region(anchors, nl, gts, None, gl))
    d['kat7_im'] = _np(head.get_im_region(anchors, gts, mode='finegrained'))
    # KAT7b: same tiny grid, tie-free float GTs (usable CPU-vs-GPU; KAT7's
    # grid-aligned GTs have centre-distance ties whose order inside
    # torch.topk is a libstdc++ nth_element/partial_sort artefact)
    gts = torch.tensor([[10.3, 12.7, 40.9, 44.2], [30.1, 5.6, 60.7, 30.4],
                        [3.3, 2.2, 9.1, 11.8]])
    gl = torch.tensor([3, 17, 42])
    ar = assigner.assign(anchors, nl, gts, None, gl)
    d['kat7b_gt'], d['kat7b_labels'] = _np(gts), _np(gl)
    d['kat7b_gt_inds'] = _np(ar.gt_inds)
    d['kat7b_max_overlaps'] = _np(ar.max_overlaps)
    d['kat7b_vlr'] = _np(assigner.get_vlr_region(anchors, nl, gts, None, gl))
    d['kat7b_im'] = _np(head.get_im_region(anchors, gts, mode='finegrained'))

    for name, pad, img_shape, num_gt, seed in TARGET_CASES:
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt),
            img_shape=img_shape,
            pad_shape=pad,
            num_gt=num_gt,
            seed=seed)
        sizes = synthetic.level_shapes(pad)
        t0 = time.time()
        (anchors_l, labels_l, lw_l, bt_l, bw_l, num_pos, num_neg, vlr_l,
         im_l) = _targets_reference(head, sizes, batch)
        labels = torch.cat(labels_l, 1)  # (N, A)
        lw = torch.cat(lw_l, 1)
        bt = torch.cat(bt_l, 1)
        vlr = torch.cat(vlr_l, 1)
        im = torch.cat(im_l, 1)
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [seed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_num_total_pos'] = np.array(num_pos)
        for n in range(labels.shape[0]):
            pos = ((labels[n] >= 0) & (labels[n] < 80)).nonzero().squeeze(1)
            d[f'{name}_{n}_pos_inds'] = _np(pos).astype(np.int32)
            d[f'{name}_{n}_pos_labels'] = _np(labels[n][pos]).astype(np.int32)
            d[f'{name}_{n}_pos_bbox_targets'] = _np(bt[n][pos])
            d[f'{name}_{n}_lw_zero_inds'] = _np(
                (lw[n] == 0).nonzero().squeeze(1)).astype(np.int32)
            vi = (vlr[n] > 0).nonzero().squeeze(1)
            d[f'{name}_{n}_vlr_inds'] = _np(vi).astype(np.int32)
            d[f'{name}_{n}_vlr_vals'] = _np(vlr[n][vi])
            d[f'{name}_{n}_im_inds'] = _np(
                (im[n] > 0).nonzero().squeeze(1)).astype(np.int32)
        print(f'  targets {name}: {time.time() - t0:.2f}s  num_total_pos='
              f'{num_pos}')
    np.savez_compressed(os.path.join(OUT, 'targets.npz'), **d)
    print('targets.npz')


# ------------------------------------------------------------ lossblock ----
LOSSBLOCK_CASES = [
    # name, pad_shape, img_shape, num_gt, batch seed, head-input seed, store grads
    ('small', (160, 224), (160, 224), [3, 1], 11, 101, True),
    ('small_crowd', (160, 224), (150, 200), [20, 7], 12, 102, True),
    ('c2', (800, 1344), (800, 1333), [7, 7], 1234, 103, False),
    ('c2_crowd', (800, 1344), (800, 1333), [1, 40], 15, 104, False),
]


def gen_lossblock():
    head = _ld_head()
    d = {}
    for name, pad, img_shape, num_gt, bseed, hseed, store in LOSSBLOCK_CASES:
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt),
            img_shape=img_shape,
            pad_shape=pad,
            num_gt=num_gt,
            seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed)
        for k in ('cls', 'reg', 'x'):
            for t in hi[k]:
                t.requires_grad_(True)
        t0 = time.time()
        losses = head.loss(hi['cls'], hi['reg'], batch['gt_bboxes'],
                           batch['gt_labels'], (hi['t_cls'], hi['t_reg']),
                           hi['x'], hi['t_x'], batch['img_metas'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        total = sum(sum(v) for v in losses.values())
        total.backward()
        d[name + '_cfg'] = np.array(
            list(pad) + list(img_shape) + [bseed, hseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)  # (8 keys, 5 levels)
        for k in ('cls', 'reg', 'x'):
            gs = [
                t.grad if t.grad is not None else torch.zeros_like(t)
                for t in hi[k]
            ]
            d[f'{name}_g{k}_abs_sum'] = np.array(
                [float(g.double().abs().sum()) for g in gs])
            d[f'{name}_g{k}_sum'] = np.array(
                [float(g.double().sum()) for g in gs])
            if store:
                for l, g in enumerate(gs):
                    d[f'{name}_g{k}_{l}'] = _np(g)
            else:  # deterministic sparse sample of the full-size gradient
                for l, g in enumerate(gs):
                    flat = _np(g).reshape(-1)
                    idx = np.arange(0, flat.size, 1009)
                    d[f'{name}_g{k}_{l}_sample'] = flat[idx]
        print(f'  lossblock {name}: {time.time() - t0:.2f}s  total='
              f'{float(total):.6f}')
    np.savez_compressed(os.path.join(OUT, 'lossblock.npz'), **d)
    print('lossblock.npz')


# ------------------------------------------------------------------ e2e ----
E2E_CASES = [
    # name, student cfg, pad, img_shape, num_gt, batch seed
    ('tiny_r18', 'configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py', (128, 160),
     (128, 150), [3, 2], 21),
    ('small_r50', 'configs/ld/ld_r50_gflv1_r101_fpn_coco_1x.py', (256, 320),
     (256, 320), [5, 2], 22),
    ('c1_r18', 'configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py', (800, 800),
     (800, 800), [7, 7], 13),
    ('c2_r50', 'configs/ld/ld_r50_gflv1_r101_fpn_coco_1x.py', (800, 1344),
     (800, 1333), [7, 7], 1234),
]


def build_reference_detector(cfg_path, imitation_method=None,
                             teacher_backbone=None):
    import mmcv
    from mmdet.models import build_detector
    cwd = os.getcwd()
    os.chdir(ref_shim.REFERENCE_ROOT)  # teacher_config is a relative path
    try:
        cfg = mmcv.Config.fromfile(cfg_path)
        m = dict(cfg.model)
        m['pretrained'] = None
        m['teacher_ckpt'] = None
        t = mmcv.Config.fromfile(m['teacher_config'])
        tm = dict(t.model)
        tm['pretrained'] = None
        if teacher_backbone is not None:
            tm['backbone'] = dict(teacher_backbone)
        m['teacher_config'] = {'model': tm}
        if imitation_method is not None:
            m['bbox_head'] = dict(m['bbox_head'])
            m['bbox_head']['imitation_method'] = imitation_method
            m['bbox_head'].setdefault('loss_im',
                                      dict(type='IMLoss', loss_weight=2.0))
        det = build_detector(m)
    finally:
        os.chdir(cwd)
    return det


def gen_e2e(cases=None):
    d = {}
    path = os.path.join(OUT, 'e2e.npz')
    if os.path.exists(path) and cases:
        d = dict(np.load(path))
    for name, cfg_path, pad, img_shape, num_gt, bseed in E2E_CASES:
        if cases and name not in cases:
            continue
        torch.manual_seed(0)
        # config 1 as shipped has imitation_method='gibox' (CUDA-only, weight
        # 0) -> evaluate it as 'finegrained' with weight 0,
        # numerically identical (0 * finite).
        method = 'finegrained'
        det = build_reference_detector(cfg_path, imitation_method=method)
        if 'r18' in name:
            det.bbox_head.loss_im.loss_weight = 0
        det.load_state_dict(
            synthetic.seeded_state_dict(det.state_dict(), seed=1))
        det.teacher_model.load_state_dict(
            synthetic.seeded_state_dict(
                det.teacher_model.state_dict(), seed=2))
        det.train()
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt),
            img_shape=img_shape,
            pad_shape=pad,
            num_gt=num_gt,
            seed=bseed)
        t0 = time.time()
        losses = det.forward_train(batch['img'], batch['img_metas'],
                                   batch['gt_bboxes'], batch['gt_labels'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        loss, log_vars = det._parse_losses(losses)
        t1 = time.time()
        loss.backward()
        t2 = time.time()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        d[name + '_log_vars'] = np.array(
            [log_vars[k] for k in LOSS_KEYS + ['loss']], dtype=np.float64)
        # a few parameter-gradient fingerprints (L2 norms), key names are part
        # of the contract
        names, norms = [], []
        for k, p in det.named_parameters():
            if p.grad is not None:
                names.append(k)
                norms.append(float(p.grad.double().norm()))
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_norms'] = np.array(norms)
        # two random projections per gradient (synthetic.grad_probe): sensitive
        # to sign, order and layout, which a norm is not
        params = dict(det.named_parameters())
        proj = []
        for k in names:
            gflat = params[k].grad.double().reshape(-1).numpy()
            proj.append([float(gflat @ synthetic.grad_probe(gflat.size, sd))
                         for sd in (0, 1)])
        d[name + '_grad_proj'] = np.array(proj)
        d[name + '_num_trainable'] = np.array(
            sum(p.numel() for p in det.parameters() if p.requires_grad))
        # state_dict contract: key names + shapes, student and teacher
        for tag, mod in (('student', det), ('teacher', det.teacher_model)):
            sd = mod.state_dict()
            d[f'{name}_{tag}_keys'] = np.array(list(sd.keys()))
            d[f'{name}_{tag}_shapes'] = np.array(
                ['x'.join(str(v) for v in t.shape) for t in sd.values()])
            d[f'{name}_{tag}_trainable'] = np.array(
                [k for k, p in mod.named_parameters() if p.requires_grad])
        # feature fingerprints
        with torch.no_grad():
            x = det.extract_feat(batch['img'])
            d[name + '_feat_abs_mean'] = np.array(
                [float(f.double().abs().mean()) for f in x])
            cls, reg = det.bbox_head(x)
            d[name + '_cls_abs_mean'] = np.array(
                [float(f.double().abs().mean()) for f in cls])
            d[name + '_reg_abs_mean'] = np.array(
                [float(f.double().abs().mean()) for f in reg])
        print(f'  e2e {name}: fwd {t1 - t0:.1f}s bwd {t2 - t1:.1f}s',
              {k: round(v, 6) for k, v in log_vars.items()})
    np.savez_compressed(path, **d)
    print('e2e.npz')


# ------------------------------------------------------- GFLv2 / LDv2 (R-V2) --
def _ldv2_head(imitation_method='finegrained'):
    """LDv2Head as configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py:27-57 builds it
    (imitation 'finegrained': 'gibox' is CUDA-only in the reference, quirk Q3)."""
    from mmdet.models import build_head
    cfg = dict(
        type='LDv2Head', num_classes=80, in_channels=256, stacked_convs=4,
        feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0],
                              octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=False, beta=2.0,
                      loss_weight=1.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
        reg_topk=4, reg_channels=64, add_mean=True,
        loss_ld=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=0.25,
                     T=10),
        reg_max=16,
        loss_kd=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=10,
                     T=2),
        loss_im=dict(type='IMLoss', loss_weight=2),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        imitation_method=imitation_method,
        train_cfg=ref_shim.ConfigDict(
            assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1,
            pos_weight=-1, debug=False),
        test_cfg=ref_shim.ConfigDict(
            nms_pre=1000, min_bbox_size=0, score_thr=0.05,
            nms=dict(type='nms', iou_threshold=0.6), max_per_img=100))
    return build_head(cfg)


def _v2_quality_tail(head, cls_feat, bbox_pred):
    """The tail of GFocalHead.forward_single (gfocal_head.py:201-217) applied
    to given tower outputs, through the head's own reg_conf modules."""
    import torch.nn.functional as F
    N, C, H, W = bbox_pred.size()
    prob = F.softmax(bbox_pred.reshape(N, 4, head.reg_max + 1, H, W), dim=2)
    prob_topk, _ = prob.topk(head.reg_topk, dim=2)
    stat = torch.cat([prob_topk, prob_topk.mean(dim=2, keepdim=True)], dim=2) \
        if head.add_mean else prob_topk
    quality_score = head.reg_conf(stat.reshape(N, -1, H, W))
    return cls_feat.sigmoid() * quality_score, quality_score


LOSSBLOCK_V2_CASES = [
    # name, pad_shape, img_shape, num_gt, batch seed, head-input seed, store grads
    ('v2_small', (160, 224), (160, 224), [3, 1], 11, 201, True),
    ('v2_small_crowd', (160, 224), (150, 200), [20, 7], 12, 202, True),
    ('v2_c2', (800, 1344), (800, 1333), [7, 7], 1234, 203, False),
]


def gen_lossblock_v2():
    head = _ldv2_head()
    head.load_state_dict(synthetic.seeded_state_dict(head.state_dict(), seed=5))
    conf = [p for n, p in head.named_parameters() if n.startswith('reg_conf')]
    d = {}
    d['reg_conf_keys'] = np.array(
        [n for n, _ in head.named_parameters() if n.startswith('reg_conf')])
    for name, pad, img_shape, num_gt, bseed, hseed, store in LOSSBLOCK_V2_CASES:
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                             num_classes=81)
        for k in ('cls', 'reg', 'x'):
            for t in hi[k]:
                t.requires_grad_(True)
        for p in conf:
            p.grad = None
        t0 = time.time()
        tails = [_v2_quality_tail(head, c, r)
                 for c, r in zip(hi['cls'], hi['reg'])]
        cls_scores = [t[0] for t in tails]
        losses = head.loss(cls_scores, hi['reg'], hi['cls'],
                           batch['gt_bboxes'], batch['gt_labels'],
                           (None, hi['t_reg'], hi['t_cls']), hi['x'],
                           hi['t_x'], batch['img_metas'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        total = sum(sum(v) for v in losses.values())
        total.backward()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed, hseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        d[name + '_quality_abs_mean'] = np.array(
            [float(t[1].double().abs().mean()) for t in tails])
        d[name + '_cls_score_abs_sum'] = np.array(
            [float(t[0].double().abs().sum()) for t in tails])
        if store:
            for l, t in enumerate(tails):
                d[f'{name}_quality_{l}'] = _np(t[1])
        for n, p in head.named_parameters():
            if n.startswith('reg_conf'):
                d[f'{name}_gparam_{n}'] = _np(p.grad)
        for k in ('cls', 'reg', 'x'):
            gs = [t.grad if t.grad is not None else torch.zeros_like(t)
                  for t in hi[k]]
            d[f'{name}_g{k}_abs_sum'] = np.array(
                [float(g.double().abs().sum()) for g in gs])
            d[f'{name}_g{k}_sum'] = np.array(
                [float(g.double().sum()) for g in gs])
            if store:
                for l, g in enumerate(gs):
                    d[f'{name}_g{k}_{l}'] = _np(g)
            else:
                for l, g in enumerate(gs):
                    flat = _np(g).reshape(-1)
                    d[f'{name}_g{k}_{l}_sample'] = flat[np.arange(0, flat.size,
                                                                  1009)]
        print(f'  lossblock {name}: {time.time() - t0:.2f}s  total='
              f'{float(total):.6f}')
    np.savez_compressed(os.path.join(OUT, 'lossblock_v2.npz'), **d)
    print('lossblock_v2.npz')


E2E_V2_CASES = [
    ('v2_tiny_r50', 'configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py', (128, 160),
     (128, 150), [3, 2], 21),
    ('v2_small_r50', 'configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py', (256, 320),
     (256, 320), [5, 2], 22),
]
# round 3 (their own file, e2e_v2_r3.npz, so the round-2 fixture stays byte-
# identical): the LDv2 step at the BASELINE config-2 size, and BASELINE config
# 5's composition "R50 <- X101, finegrained" (no such config file
# exists in the reference; the student config's teacher backbone is replaced by
# configs/imv2/gflv2_x101_fpn_2x_coco.py:8-20's ResNeXt-101 32x4d settings,
# WITHOUT its DCN -- mmcv's compiled op is absent here, parity of DCN stays
# unpinned)
E2E_V2_R3_CASES = [
    ('v2_c2_r50', 'configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py', (800, 1344),
     (800, 1333), [7, 7], 1234, None),
    ('v2x_small_r50', 'configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py', (256, 320),
     (256, 320), [5, 2], 22, 'x101'),
    ('v2x_c2_r50', 'configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py', (800, 1344),
     (800, 1333), [7, 7], 1234, 'x101'),
]
X101_BACKBONE = dict(type='ResNeXt', depth=101, groups=32, base_width=4,
                     num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                     norm_cfg=dict(type='BN', requires_grad=True),
                     norm_eval=True, style='pytorch')


def gen_e2e_v2():
    """Whole-model LDv2 step (kd_one_stage.py:46-81 with LDv2Head / a
    GFocalHead teacher) from the reference's own config file."""
    d = {}
    for name, cfg_path, pad, img_shape, num_gt, bseed in E2E_V2_CASES:
        torch.manual_seed(0)
        det = build_reference_detector(cfg_path, imitation_method='finegrained')
        det.load_state_dict(
            synthetic.seeded_state_dict(det.state_dict(), seed=1))
        det.teacher_model.load_state_dict(
            synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2))
        det.train()
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        t0 = time.time()
        losses = det.forward_train(batch['img'], batch['img_metas'],
                                   batch['gt_bboxes'], batch['gt_labels'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        loss, log_vars = det._parse_losses(losses)
        loss.backward()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        d[name + '_log_vars'] = np.array(
            [log_vars[k] for k in LOSS_KEYS + ['loss']], dtype=np.float64)
        names, norms = [], []
        for k, p in det.named_parameters():
            if p.grad is not None:
                names.append(k)
                norms.append(float(p.grad.double().norm()))
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_norms'] = np.array(norms)
        for tag, mod in (('student', det), ('teacher', det.teacher_model)):
            sd = mod.state_dict()
            d[f'{name}_{tag}_keys'] = np.array(list(sd.keys()))
            d[f'{name}_{tag}_shapes'] = np.array(
                ['x'.join(str(v) for v in t.shape) for t in sd.values()])
            d[f'{name}_{tag}_trainable'] = np.array(
                [k for k, p in mod.named_parameters() if p.requires_grad])
        with torch.no_grad():
            x = det.extract_feat(batch['img'])
            cls, reg, feat = det.bbox_head(x)
            d[name + '_cls_score_abs_mean'] = np.array(
                [float(f.double().abs().mean()) for f in cls])
            d[name + '_cls_feat_abs_mean'] = np.array(
                [float(f.double().abs().mean()) for f in feat])
        print(f'  e2e {name}: {time.time() - t0:.1f}s',
              {k: round(v, 6) for k, v in log_vars.items()})
    np.savez_compressed(os.path.join(OUT, 'e2e_v2.npz'), **d)
    print('e2e_v2.npz')


def gen_e2e_v2_r3(cases=None):
    """gen_e2e_v2's record plus two random projections per parameter gradient
    (synthetic.grad_probe, as e2e.npz has them), for the round-3 cases."""
    d = {}
    path = os.path.join(OUT, 'e2e_v2_r3.npz')
    if os.path.exists(path) and cases:
        d = dict(np.load(path))
    for name, cfg_path, pad, img_shape, num_gt, bseed, teacher in \
            E2E_V2_R3_CASES:
        if cases and name not in cases:
            continue
        torch.manual_seed(0)
        det = build_reference_detector(
            cfg_path, imitation_method='finegrained',
            teacher_backbone=X101_BACKBONE if teacher == 'x101' else None)
        det.load_state_dict(
            synthetic.seeded_state_dict(det.state_dict(), seed=1))
        det.teacher_model.load_state_dict(
            synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2))
        det.train()
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        t0 = time.time()
        losses = det.forward_train(batch['img'], batch['img_metas'],
                                   batch['gt_bboxes'], batch['gt_labels'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        loss, log_vars = det._parse_losses(losses)
        loss.backward()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        d[name + '_log_vars'] = np.array(
            [log_vars[k] for k in LOSS_KEYS + ['loss']], dtype=np.float64)
        names, norms, proj = [], [], []
        for k, p in det.named_parameters():
            if p.grad is not None:
                names.append(k)
                norms.append(float(p.grad.double().norm()))
                gflat = p.grad.double().reshape(-1).numpy()
                proj.append([float(gflat @ synthetic.grad_probe(gflat.size, sd))
                             for sd in (0, 1)])
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_norms'] = np.array(norms)
        d[name + '_grad_proj'] = np.array(proj)
        sd = det.teacher_model.state_dict()
        d[name + '_teacher_keys'] = np.array(list(sd.keys()))
        d[name + '_teacher_shapes'] = np.array(
            ['x'.join(str(v) for v in t.shape) for t in sd.values()])
        with torch.no_grad():
            tx = det.teacher_model.extract_feat(batch['img'])
            d[name + '_teacher_feat_abs_mean'] = np.array(
                [float(f.double().abs().mean()) for f in tx])
        print(f'  e2e {name}: {time.time() - t0:.1f}s',
              {k: round(v, 6) for k, v in log_vars.items()})
    np.savez_compressed(path, **d)
    print('e2e_v2_r3.npz')


def _two_rank_worker(rank, world, port, cases, ret):
    """One rank of the REFERENCE under a real gloo group: LDHead.loss on this
    rank's case (its own reduce_mean all-reduces, ld_head.py:338-341,362-365) and
    BaseDetector._parse_losses (base.py:185-218) with its per-key all-reduce."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mmdet.models.detectors.base import BaseDetector
        name, pad, img_shape, num_gt, bseed, hseed, _ = \
            [c for c in LOSSBLOCK_CASES if c[0] == cases[rank]][0]
        head = _ld_head()
        batch = synthetic.synthetic_batch(num_imgs=len(num_gt), img_shape=img_shape,
                                          pad_shape=pad, num_gt=num_gt, seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed)
        losses = head.loss(hi['cls'], hi['reg'], batch['gt_bboxes'],
                           batch['gt_labels'], (hi['t_cls'], hi['t_reg']),
                           hi['x'], hi['t_x'], batch['img_metas'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        loss, log_vars = BaseDetector._parse_losses(None, losses)
        ret[rank] = dict(table=table, loss=float(loss.detach()),
                         log_vars=[log_vars[k] for k in LOSS_KEYS + ['loss']])
    finally:
        dist.destroy_process_group()


def gen_lossblock_2rank():
    """Round 4: the reference's own cross-rank arithmetic.
    Two gloo ranks execute LDHead.loss on DIFFERENT lossblock cases; the tables
    each rank gets (QFL divided by max(mean_ranks(num_total_pos), 1), GIoU / DFL
    by mean_ranks(sum weight_targets + 1e-6)) and the rank-averaged log_vars are
    the fixture (tests/golden/lossblock_2rank.npz)."""
    import socket
    import torch.multiprocessing as mp
    d = {}
    for tag, cases in (('small_pair', ('small', 'small_crowd')),
                       ('c2_pair', ('c2', 'c2_crowd'))):
        so = socket.socket()
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
        so.close()
        ctx = mp.get_context('spawn')
        ret = ctx.Manager().dict()
        procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, cases, ret))
                 for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        d[tag + '_cases'] = np.array(cases)
        for r in range(2):
            d[f'{tag}_r{r}_losses'] = ret[r]['table'].astype(np.float64)
            d[f'{tag}_r{r}_loss'] = np.array(ret[r]['loss'])
            d[f'{tag}_r{r}_log_vars'] = np.array(ret[r]['log_vars'], dtype=np.float64)
        print(f'  2-rank {tag}: loss', ret[0]['loss'], ret[1]['loss'], 'log loss',
              ret[0]['log_vars'][-1])
    np.savez_compressed(os.path.join(OUT, 'lossblock_2rank.npz'), **d)
    print('lossblock_2rank.npz')


def gen_grad_samples(cases=None):
    """Round 4: whole-step parameter gradients of the
    REFERENCE at the BASELINE config-2 size, ELEMENT-WISE -- for every trainable
    parameter 256 sampled elements (synthetic.grad_sample_idx: a fixed integer
    hash of the element count) and the gradient's max |g|.  Their own file
    (grad_samples.npz): the earlier fixtures stay byte-identical.  The step is
    exactly gen_e2e's / gen_e2e_v2_r3's (same seeds, same batch)."""
    d = {}
    path = os.path.join(OUT, 'grad_samples.npz')
    if os.path.exists(path) and cases:
        d = dict(np.load(path))
    todo = [(n, c, p, i, g, b, None) for n, c, p, i, g, b in E2E_CASES
            if n in ('small_r50', 'c2_r50')]
    todo += [c for c in E2E_V2_R3_CASES if c[0] == 'v2_c2_r50']
    for name, cfg_path, pad, img_shape, num_gt, bseed, teacher in todo:
        if cases and name not in cases:
            continue
        torch.manual_seed(0)
        det = build_reference_detector(cfg_path, imitation_method='finegrained')
        det.load_state_dict(
            synthetic.seeded_state_dict(det.state_dict(), seed=1))
        det.teacher_model.load_state_dict(
            synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2))
        det.train()
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        t0 = time.time()
        losses = det.forward_train(batch['img'], batch['img_metas'],
                                   batch['gt_bboxes'], batch['gt_labels'])
        loss, log_vars = det._parse_losses(losses)
        loss.backward()
        names, vals, amax = [], [], []
        for k, p in det.named_parameters():
            if p.grad is None:
                continue
            gflat = p.grad.reshape(-1)
            idx = synthetic.grad_sample_idx(gflat.numel())
            v = np.zeros(256, dtype=np.float32)
            v[:idx.size] = gflat[torch.from_numpy(idx)].numpy()
            names.append(k)
            vals.append(v)
            amax.append(float(gflat.abs().max()))
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_samples'] = np.stack(vals)
        d[name + '_grad_absmax'] = np.array(amax)
        d[name + '_loss'] = np.array(float(loss.detach()))
        print(f'  grad samples {name}: {time.time() - t0:.1f}s, {len(names)} '
              f'parameters, loss {float(loss.detach()):.6f}')
    np.savez_compressed(path, **d)
    print('grad_samples.npz')


def gen_grad_truth64(cases=('small_r50', 'c2_r50')):
    """How well fp32 DEFINES these gradients (round 4): the same steps with the
    nets evaluated in float64 (oracle/net_oracle.py on float64 state_dicts and
    images; the loss block stays the fp32 numpy oracle), sampled at the elements
    of grad_samples.npz, plus the REFERENCE's own deviation from it per
    parameter.  On small_r50 the reference's fp32 CPU gradients differ from the
    float64 evaluation by up to 4e-4 of max|g| (median 4e-5): that, not 1e-5, is
    the floor any fp32 implementation can be held to element-wise.  This fixture
    is produced by the ORACLE, not by the reference (the reference is fp32-only
    here); the oracle is pinned on the reference element-wise in
    tests/test_oracle_golden.py."""
    import net_oracle as NO
    from ld_amd import build_detector, model_zoo
    gs = np.load(os.path.join(OUT, 'grad_samples.npz'))
    ge = np.load(os.path.join(OUT, 'e2e.npz'))
    d = {}
    for name in cases:
        cfg = ge[name + '_cfg']
        pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
        num_gt = [int(x) for x in ge[name + '_num_gt']]
        batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
        det = build_detector(model_zoo.ld_detector(50, 101))
        ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
        tsd = synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2)
        to64 = lambda sd: {k: (v.double() if v.is_floating_point() else v)  # noqa: E731
                           for k, v in sd.items()}
        b64 = dict(batch)
        b64['img'] = batch['img'].double()
        t0 = time.time()
        res = NO.ld_train_step(to64(ssd), to64(tsd), b64, 50, 101, None)
        names = [str(k) for k in gs[name + '_grad_names']]
        vals, referr = [], []
        for k, ref, am in zip(names, gs[name + '_grad_samples'],
                              gs[name + '_grad_absmax']):
            flat = res['grads'][k].reshape(-1)
            idx = synthetic.grad_sample_idx(flat.numel())
            v = np.zeros(256, dtype=np.float64)
            v[:idx.size] = flat[torch.from_numpy(idx)].double().numpy()
            vals.append(v)
            referr.append(float(np.abs(ref[:idx.size] - v[:idx.size]).max()))
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_truth64'] = np.stack(vals)
        d[name + '_ref_abs_err'] = np.array(referr)
        rel = np.array(referr) / np.maximum(gs[name + '_grad_absmax'], 1e-30)
        print(f'  truth64 {name}: {time.time() - t0:.1f}s; reference vs float64, '
              f'max|err| / max|g| per parameter: median {np.median(rel):.2e}, '
              f'max {rel.max():.2e}')
    np.savez_compressed(os.path.join(OUT, 'grad_truth64.npz'), **d)
    print('grad_truth64.npz')


RESNEXT_CASES = [
    # name, depth, input (N, H, W), seed, sample step of the stored features
    ('x101_small', 101, (2, 64, 96), 31, 1),
    ('x101_mid', 101, (1, 224, 320), 32, 37),
    ('x50_odd', 50, (1, 75, 101), 33, 1),
]


def gen_resnext():
    """The reference's ResNeXt (mmdet/models/backbones/resnext.py:11-153, pure
    torch: grouped nn.Conv2d) on seeded weights and inputs: the four stage
    outputs, element-wise (every `step`-th element of the flattened map)."""
    from mmdet.models.backbones import ResNeXt
    d = {}
    for name, depth, (n, h, w), seed, step in RESNEXT_CASES:
        cfg = dict(X101_BACKBONE)
        cfg.pop('type')
        cfg['depth'] = depth
        net = ResNeXt(**cfg)
        net.load_state_dict(synthetic.seeded_state_dict(net.state_dict(),
                                                        seed=seed))
        net.eval()
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.randn(n, 3, h, w, generator=g)
        with torch.no_grad():
            outs = net(x)
        d[name + '_cfg'] = np.array([depth, n, h, w, seed, step])
        sd = net.state_dict()
        d[name + '_keys'] = np.array(list(sd.keys()))
        d[name + '_shapes'] = np.array(
            ['x'.join(str(v) for v in t.shape) for t in sd.values()])
        for i, o in enumerate(outs):
            d[f'{name}_shape{i}'] = np.array(o.shape)
            d[f'{name}_out{i}'] = _np(o).reshape(-1)[::step].astype(np.float32)
            d[f'{name}_absmean{i}'] = np.array(float(o.double().abs().mean()))
        print('  resnext', name, [tuple(o.shape) for o in outs])
    np.savez_compressed(os.path.join(OUT, 'resnext.npz'), **d)
    print('resnext.npz')


# ------------------------------------------- other imitation regions (8f-4) --
def _patch_for_gibox():
    """get_gi_region is CUDA-only in the reference (quirk Q3): it calls
    ``torch.arange(...).cuda()`` (ld_head.py:631) and
    ``torch.ops.torchvision.nms`` (:637).  To execute it here, ``.cuda()``
    becomes the identity and the torchvision op is registered with its
    published semantics (greedy, descending score, IoU > thr suppresses;
    restated -- the compiled op is absent), so everything around the NMS core
    is the reference's own arithmetic."""
    torch.Tensor.cuda = lambda self, *a, **k: self

    def nms(dets, scores, iou_threshold):
        order = torch.argsort(scores, descending=True, stable=True)
        b = dets[order]
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        keep, alive = [], torch.ones(len(order), dtype=torch.bool)
        for i in range(len(order)):
            if not alive[i]:
                continue
            keep.append(int(order[i]))
            if len(keep) >= 64:  # callers take [:10]
                break
            w = (torch.minimum(b[i, 2], b[i + 1:, 2]) -
                 torch.maximum(b[i, 0], b[i + 1:, 0])).clamp(min=0)
            h = (torch.minimum(b[i, 3], b[i + 1:, 3]) -
                 torch.maximum(b[i, 1], b[i + 1:, 1])).clamp(min=0)
            inter = w * h
            iou = inter / (area[i] + area[i + 1:] - inter)
            alive[i + 1:] &= ~(iou > iou_threshold)
        return torch.tensor(keep, dtype=torch.long)

    try:
        torch.library.define('torchvision::nms',
                             '(Tensor dets, Tensor scores, float iou_threshold) '
                             '-> Tensor')
        torch.library.impl('torchvision::nms', 'cpu')(nms)
    except RuntimeError:
        pass  # already registered in this process


IMITATION_CASES = [
    # name, head kind, method, pad, img_shape, num_gt, batch seed, input seed
    ('fitnet_small', 'v1', 'fitnet', (160, 224), (150, 200), [20, 7], 12, 301),
    ('fitnet_c2', 'v1', 'fitnet', (800, 1344), (800, 1333), [7, 7], 1234, 302),
    ('gibox_small', 'v1', 'gibox', (160, 224), (160, 224), [3, 1], 11, 303),
    ('gibox_c2', 'v1', 'gibox', (800, 1344), (800, 1333), [7, 7], 1234, 304),
    ('gibox_v2_small', 'v2', 'gibox', (160, 224), (150, 200), [20, 7], 12, 305),
]


def gen_imitation():
    _patch_for_gibox()
    d = {}
    for name, kind, method, pad, img_shape, num_gt, bseed, hseed in \
            IMITATION_CASES:
        head = _ld_head(method) if kind == 'v1' else _ldv2_head(method)
        if kind == 'v2':
            head.load_state_dict(
                synthetic.seeded_state_dict(head.state_dict(), seed=5))
        picked = []
        if method == 'gibox':
            orig = head.get_gi_region

            def rec(*a, _o=orig, **k):
                out = _o(*a, **k)
                picked.append(_np(out).astype(np.int64))
                return out

            head.get_gi_region = rec
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(
            len(num_gt), sizes, seed=hseed,
            num_classes=80 if kind == 'v1' else 81)
        for k in ('cls', 'reg', 'x'):
            for t in hi[k]:
                t.requires_grad_(True)
        if kind == 'v1':
            losses = head.loss(hi['cls'], hi['reg'], batch['gt_bboxes'],
                               batch['gt_labels'], (hi['t_cls'], hi['t_reg']),
                               hi['x'], hi['t_x'], batch['img_metas'])
        else:
            scores = [_v2_quality_tail(head, c, r)[0]
                      for c, r in zip(hi['cls'], hi['reg'])]
            losses = head.loss(scores, hi['reg'], hi['cls'],
                               batch['gt_bboxes'], batch['gt_labels'],
                               (None, hi['t_reg'], hi['t_cls']), hi['x'],
                               hi['t_x'], batch['img_metas'])
        table = np.stack(
            [np.array([float(v.detach()) for v in losses[k]]) for k in LOSS_KEYS])
        sum(sum(v) for v in losses.values()).backward()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed, hseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        for l, idx in enumerate(picked):
            d[f'{name}_gi_idx_{l}'] = idx
        gx = [t.grad if t.grad is not None else torch.zeros_like(t)
              for t in hi['x']]
        d[name + '_gx_abs_sum'] = np.array(
            [float(g.double().abs().sum()) for g in gx])
        d[name + '_gx_nonzero_rows'] = np.array(
            [int((g.abs().sum(1) > 0).sum()) for g in gx])
        print(f'  imitation {name}: loss_im', table[7])
    np.savez_compressed(os.path.join(OUT, 'imitation.npz'), **d)
    print('imitation.npz')


# ------------------------------------------------------------ inference ----
# GFLHead.get_bboxes (anchor_head.py:497-589 -> gfl_head.py:354-451 ->
# post_processing/bbox_nms.py:70-195 multiclass_nms -> mmcv.ops.batched_nms,
# the latter restated in ref_shim).  Head outputs come from
# synthetic.synthetic_head_inputs; the class logits are re-scaled to
# cls * cls_scale + cls_shift so the candidate count lands on either side of
# batched_nms' split_thr = 10000.
# the cases and their seeded inputs live in ld_amd.synthetic (INFER_CASES,
# infer_inputs) so the tests regenerate exactly the same tensors


def gen_infer():
    import mmcv
    head = _ld_head()
    d = {}
    for case in synthetic.INFER_CASES:
        name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
        cls, reg, metas = synthetic.infer_inputs(case)
        cfg = mmcv.ConfigDict(dict(nms_pre=nms_pre, min_bbox_size=0,
                                   score_thr=0.05,
                                   nms=dict(type='nms', iou_threshold=0.6),
                                   max_per_img=100))
        t0 = time.time()
        for rescale in (False, True):
            res = head.get_bboxes(cls, reg, metas, cfg=cfg, rescale=rescale)
            tag = f'{name}_r{int(rescale)}'
            for i, (db, dl) in enumerate(res):
                d[f'{tag}_bboxes_{i}'] = _np(db).astype(np.float32)
                d[f'{tag}_labels_{i}'] = _np(dl).astype(np.int64)
        # pre-NMS stage (top-k selection + decode), rescale=False
        pre = head.get_bboxes(cls, reg, metas, cfg=cfg, rescale=False,
                              with_nms=False)
        for i, (bb, sc) in enumerate(pre):
            sc = sc[:, :-1]  # drop the padded background column
            d[f'{name}_pre_count_{i}'] = np.array(bb.shape[0])
            cand = int((sc > 0.05).sum())
            d[f'{name}_candidates_{i}'] = np.array(cand)
            if store:
                d[f'{name}_pre_bboxes_{i}'] = _np(bb).astype(np.float32)
                d[f'{name}_pre_scores_{i}'] = _np(sc).astype(np.float32)
            else:  # fingerprints of the full-size stage
                d[f'{name}_pre_bboxes_sum_{i}'] = np.array(
                    float(bb.double().sum()))
                d[f'{name}_pre_scores_sum_{i}'] = np.array(
                    float(sc.double().sum()))
                d[f'{name}_pre_maxscore_{i}'] = _np(sc.max(1)[0]).astype(
                    np.float32)
        d[name + '_cfg'] = np.array(list(pad) + [seed, nms_pre])
        print(f'[infer] {name}: {time.time() - t0:.1f}s, dets',
              [int(d[f"{name}_r0_labels_{i}"].shape[0])
               for i in range(len(img_shapes))], 'candidates',
              [int(d[f"{name}_candidates_{i}"]) for i in range(len(img_shapes))])
    np.savez_compressed(os.path.join(OUT, 'infer.npz'), **d)


def gen_infer_v2():
    """GFocalHead.get_bboxes (gfocal_head.py:317-596; LDv2Head inherits it):
    no sigmoid -- the maps are cls_score = sigmoid(cls) * quality -- and 81
    score channels (use_sigmoid=False)."""
    import mmcv
    head = _ldv2_head()
    d = {}
    for case in synthetic.INFER_CASES:
        name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
        if name not in synthetic.INFER_V2_CASES:
            continue
        cls, reg, metas = synthetic.infer_inputs_prob(case)
        cfg = mmcv.ConfigDict(dict(nms_pre=nms_pre, min_bbox_size=0,
                                   score_thr=0.05,
                                   nms=dict(type='nms', iou_threshold=0.6),
                                   max_per_img=100))
        for rescale in (False, True):
            res = head.get_bboxes(cls, reg, None, metas, cfg=cfg,
                                  rescale=rescale)
            tag = f'{name}_r{int(rescale)}'
            for i, (db, dl) in enumerate(res):
                d[f'{tag}_bboxes_{i}'] = _np(db).astype(np.float32)
                d[f'{tag}_labels_{i}'] = _np(dl).astype(np.int64)
        d[name + '_cfg'] = np.array(list(pad) + [seed, nms_pre])
        print(f'[infer_v2] {name}: dets',
              [int(d[f"{name}_r0_labels_{i}"].shape[0])
               for i in range(len(img_shapes))], 'labels == 80 present:',
              [bool((d[f"{name}_r0_labels_{i}"] == 80).any())
               for i in range(len(img_shapes))])
    np.savez_compressed(os.path.join(OUT, 'infer_v2.npz'), **d)


def gen_infer_ctr():
    """ATSSGFLHead / FCOSGFLHead get_bboxes (atss_gfl_head.py:420-575,
    fcos_gfl_head.py:347-546): centerness-weighted top-k and scores
    (multiclass_nms score_factors), FCOS points for the latter."""
    import mmcv
    d = {}
    for tag, head in (('atss', _ld_atss_head()), ('fcos', _ld_fcos_head())):
        for case in synthetic.INFER_CASES:
            name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
            if name not in synthetic.INFER_V2_CASES:
                continue
            cls, reg, metas = synthetic.infer_inputs(case)
            ctr = synthetic.synthetic_centerness(
                len(img_shapes), synthetic.level_shapes(pad), seed=seed)
            cfg = mmcv.ConfigDict(dict(nms_pre=nms_pre, min_bbox_size=0,
                                       score_thr=0.05,
                                       nms=dict(type='nms', iou_threshold=0.6),
                                       max_per_img=100))
            for rescale in (False, True):
                res = head.get_bboxes(cls, reg, ctr, metas, cfg=cfg,
                                      rescale=rescale)
                key = f'{tag}_{name}_r{int(rescale)}'
                for i, (db, dl) in enumerate(res):
                    d[f'{key}_bboxes_{i}'] = _np(db).astype(np.float32)
                    d[f'{key}_labels_{i}'] = _np(dl).astype(np.int64)
            print(f'[infer_ctr] {tag} {name}: dets',
                  [int(d[f"{tag}_{name}_r0_labels_{i}"].shape[0])
                   for i in range(len(img_shapes))], 'min score',
                  [float(d[f"{tag}_{name}_r0_bboxes_{i}"][:, 4].min())
                   for i in range(len(img_shapes))])
    np.savez_compressed(os.path.join(OUT, 'infer_ctr.npz'), **d)


def gen_infer_retina():
    """RetinaGFLHead / LDRetinaHead get_bboxes (retina_gfl_head.py:301-412;
    anchor_head.py:497-589): 9 anchors per cell, top-k per level over all
    (cell, anchor) rows."""
    import mmcv
    head = _ld_retina_head()
    d = {}
    for case in synthetic.INFER_CASES:
        name, pad, img_shapes, sfs, seed, nms_pre, cs, sh, store = case
        if name not in synthetic.INFER_V2_CASES:
            continue
        cls, reg, metas = synthetic.infer_inputs_retina(case)
        cfg = mmcv.ConfigDict(dict(nms_pre=nms_pre, min_bbox_size=0,
                                   score_thr=0.05,
                                   nms=dict(type='nms', iou_threshold=0.6),
                                   max_per_img=100))
        for rescale in (False, True):
            res = head.get_bboxes(cls, reg, metas, cfg=cfg, rescale=rescale)
            key = f'{name}_r{int(rescale)}'
            for i, (db, dl) in enumerate(res):
                d[f'{key}_bboxes_{i}'] = _np(db).astype(np.float32)
                d[f'{key}_labels_{i}'] = _np(dl).astype(np.int64)
        print(f'[infer_retina] {name}: dets',
              [int(d[f"{name}_r0_labels_{i}"].shape[0])
               for i in range(len(img_shapes))])
    np.savez_compressed(os.path.join(OUT, 'infer_retina.npz'), **d)


def _ld_atss_head():
    """LDATSSHead as configs/ld/ld_r50_atss_r101_1x.py:29-58 builds it."""
    from mmdet.models import build_head
    cfg = dict(
        type='LDATSSHead', num_classes=80, in_channels=256, stacked_convs=4,
        feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0],
                              octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                        target_means=[.0, .0, .0, .0],
                        target_stds=[0.1, 0.1, 0.2, 0.2]),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        loss_ld=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=0.25,
                     T=10),
        loss_kd=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=10,
                     T=2),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True,
                             loss_weight=1.0),
        train_cfg=ref_shim.ConfigDict(
            assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1,
            pos_weight=-1, debug=False),
        test_cfg=ref_shim.ConfigDict(
            nms_pre=1000, min_bbox_size=0, score_thr=0.05,
            nms=dict(type='nms', iou_threshold=0.6), max_per_img=100))
    return build_head(cfg)


ATSS_KEYS = ['loss_cls', 'loss_bbox', 'loss_ld', 'loss_ld_neg', 'loss_cls_kd',
             'loss_centerness']


def gen_lossblock_atss():
    """LDATSSHead.loss (ld_atss.py:168-250) executed by the reference on the
    LOSSBLOCK_CASES inputs + synthetic centerness maps: loss table (6 keys x 5
    levels) and the gradients of the sum of all entries wrt cls / reg /
    centerness."""
    head = _ld_atss_head()
    d = {}
    for name, pad, img_shape, num_gt, bseed, hseed, store in LOSSBLOCK_CASES:
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed)
        hi['ctr'] = synthetic.synthetic_centerness(len(num_gt), sizes,
                                                   seed=hseed)
        for k in ('cls', 'reg', 'ctr'):
            for t in hi[k]:
                t.requires_grad_(True)
        t0 = time.time()
        losses = head.loss(hi['cls'], hi['reg'], hi['ctr'],
                           batch['gt_bboxes'], batch['gt_labels'],
                           (hi['t_cls'], hi['t_reg'], None),
                           batch['img_metas'])
        table = np.stack([np.array([float(v.detach()) for v in losses[k]])
                          for k in ATSS_KEYS])
        total = sum(sum(v) for v in losses.values())
        total.backward()
        d[name + '_cfg'] = np.array(
            list(pad) + list(img_shape) + [bseed, hseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        for k in ('cls', 'reg', 'ctr'):
            gs = [t.grad if t.grad is not None else torch.zeros_like(t)
                  for t in hi[k]]
            d[f'{name}_g{k}_abs_sum'] = np.array(
                [float(g.double().abs().sum()) for g in gs])
            d[f'{name}_g{k}_sum'] = np.array(
                [float(g.double().sum()) for g in gs])
            for l, g in enumerate(gs):
                if store:
                    d[f'{name}_g{k}_{l}'] = _np(g)
                else:
                    flat = _np(g).reshape(-1)
                    d[f'{name}_g{k}_{l}_sample'] = flat[
                        np.arange(0, flat.size, 1009)]
        print(f'  lossblock_atss {name}: {time.time() - t0:.2f}s  total='
              f'{float(total):.6f}', table.sum(1))
    np.savez_compressed(os.path.join(OUT, 'lossblock_atss.npz'), **d)
    print('lossblock_atss.npz')


def _ld_fcos_head():
    """LDFCOSHead as configs/ld/ld_r50_fcos_r101_1x.py:26-49 builds it."""
    from mmdet.models import build_head
    cfg = dict(
        type='LDFCOSHead', num_classes=80, in_channels=256, stacked_convs=4,
        feat_channels=256, strides=[8, 16, 32, 64, 128],
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=1.0),
        loss_ld=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=0.25,
                     T=10),
        loss_kd=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=10,
                     T=2),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True,
                             loss_weight=1.0),
        norm_on_bbox=False, centerness_on_reg=True, dcn_on_last_conv=False,
        center_sampling=True, conv_bias=True,
        train_cfg=ref_shim.ConfigDict(
            assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5,
                          neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1),
            allowed_border=-1, pos_weight=-1, debug=False),
        test_cfg=ref_shim.ConfigDict(
            nms_pre=1000, min_bbox_size=0, score_thr=0.05,
            nms=dict(type='nms', iou_threshold=0.6), max_per_img=100))
    return build_head(cfg)


def gen_lossblock_fcos():
    """LDFCOSHead (ld_fcos_head.py): point targets and loss tables + gradients
    executed by the reference on the LOSSBLOCK_CASES inputs."""
    head = _ld_fcos_head()
    d = {}
    for name, pad, img_shape, num_gt, bseed, hseed, store in LOSSBLOCK_CASES:
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed)
        hi['ctr'] = synthetic.synthetic_centerness(len(num_gt), sizes,
                                                   seed=hseed)
        # the reference's targets (labels with the "remain" code C + 1)
        pts = head.get_points(sizes, torch.float32, 'cpu')
        labels, bbox_targets = head.get_targets(pts, batch['gt_bboxes'],
                                                batch['gt_labels'])
        for l in range(len(sizes)):
            d[f'{name}_labels_{l}'] = _np(labels[l]).astype(np.int64)
            d[f'{name}_bbox_targets_{l}'] = _np(bbox_targets[l])
        for k in ('cls', 'reg', 'ctr'):
            for t in hi[k]:
                t.requires_grad_(True)
        losses = head.loss(hi['cls'], hi['reg'], hi['ctr'],
                           batch['gt_bboxes'], batch['gt_labels'],
                           (hi['t_cls'], hi['t_reg'], None),
                           batch['img_metas'])
        table = np.stack([np.array([float(v.detach()) for v in losses[k]])
                          for k in ATSS_KEYS])
        total = sum(sum(v) for v in losses.values())
        total.backward()
        d[name + '_cfg'] = np.array(
            list(pad) + list(img_shape) + [bseed, hseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        for k in ('cls', 'reg', 'ctr'):
            gs = [t.grad if t.grad is not None else torch.zeros_like(t)
                  for t in hi[k]]
            d[f'{name}_g{k}_abs_sum'] = np.array(
                [float(g.double().abs().sum()) for g in gs])
            for l, g in enumerate(gs):
                if store:
                    d[f'{name}_g{k}_{l}'] = _np(g)
                else:
                    flat = _np(g).reshape(-1)
                    d[f'{name}_g{k}_{l}_sample'] = flat[
                        np.arange(0, flat.size, 1009)]
        print(f'  lossblock_fcos {name}: total={float(total):.6f}',
              table.sum(1), 'pos',
              int(sum(((l_ >= 0) & (l_ < 80)).sum() for l_ in labels)),
              'remain', int(sum((l_ == 81).sum() for l_ in labels)))
    np.savez_compressed(os.path.join(OUT, 'lossblock_fcos.npz'), **d)


RETINA_KEYS = ['loss_cls', 'loss_bbox', 'loss_ld', 'loss_ld_vlr',
               'loss_cls_kd']


def _ld_retina_head():
    """LDRetinaHead as configs/ld/ld_retina_r50_1x.py:28-62 builds it."""
    from mmdet.models import build_head
    cfg = dict(
        type='LDRetinaHead', num_classes=80, in_channels=256, stacked_convs=4,
        feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4,
                              scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                              strides=[8, 16, 32, 64, 128]),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                        target_means=[.0, .0, .0, .0],
                        target_stds=[1.0, 1.0, 1.0, 1.0]),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        loss_ld=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=5,
                     T=10),
        loss_kd=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=10,
                     T=8),
        reg_decoded_bbox=True,
        train_cfg=ref_shim.ConfigDict(
            assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5,
                          neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1),
            allowed_border=-1, pos_weight=-1, debug=False),
        test_cfg=ref_shim.ConfigDict(
            nms_pre=1000, min_bbox_size=0, score_thr=0.05,
            nms=dict(type='nms', iou_threshold=0.6), max_per_img=100))
    return build_head(cfg)


def gen_lossblock_retina():
    """LDRetinaHead (ld_retina.py): MaxIoU targets + VLR region over the 9
    anchors per position, loss table (5 keys x 5 levels) and gradients,
    executed by the reference on the LOSSBLOCK_CASES inputs with 9-anchor head
    outputs."""
    head = _ld_retina_head()
    d = {}
    for name, pad, img_shape, num_gt, bseed, hseed, store in LOSSBLOCK_CASES:
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        sizes = synthetic.level_shapes(pad)
        hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                             num_anchors=9)
        t0 = time.time()
        anchor_list, valid_flag_list = head.get_anchors(
            sizes, batch['img_metas'], device='cpu')
        if name == 'small':
            for l, a in enumerate(anchor_list[0]):
                d[f'anchors_{l}'] = _np(a)
        tg = head.get_targets(anchor_list, valid_flag_list,
                              batch['gt_bboxes'], batch['img_metas'],
                              gt_labels_list=batch['gt_labels'],
                              label_channels=80)
        (labels, label_weights, bbox_targets, bbox_weights, num_pos, num_neg,
         vlr) = tg
        d[name + '_num_total_pos'] = np.array(num_pos)
        for l in range(len(sizes)):
            d[f'{name}_labels_{l}'] = _np(labels[l]).astype(np.int64)
            d[f'{name}_label_weights_{l}'] = _np(label_weights[l])
            d[f'{name}_vlr_{l}'] = _np(vlr[l])
            bw = _np(bbox_weights[l])
            d[f'{name}_bbox_pos_{l}'] = _np(bbox_targets[l])[bw[..., 0] > 0]
        for k in ('cls', 'reg'):
            for t in hi[k]:
                t.requires_grad_(True)
        losses = head.loss(hi['cls'], hi['reg'], batch['gt_bboxes'],
                           batch['gt_labels'], (hi['t_cls'], hi['t_reg']),
                           batch['img_metas'])
        table = np.stack([np.array([float(v.detach()) for v in losses[k]])
                          for k in RETINA_KEYS])
        total = sum(sum(v) for v in losses.values())
        total.backward()
        d[name + '_cfg'] = np.array(
            list(pad) + list(img_shape) + [bseed, hseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        for k in ('cls', 'reg'):
            gs = [t.grad if t.grad is not None else torch.zeros_like(t)
                  for t in hi[k]]
            d[f'{name}_g{k}_abs_sum'] = np.array(
                [float(g.double().abs().sum()) for g in gs])
            d[f'{name}_g{k}_sum'] = np.array(
                [float(g.double().sum()) for g in gs])
            for l, g in enumerate(gs):
                # 1 332 channels: sampled even for the small cases (every 11th
                # element; every 1009th at the C2 size)
                flat = _np(g).reshape(-1)
                d[f'{name}_g{k}_{l}_sample'] = flat[
                    np.arange(0, flat.size, 11 if store else 1009)]
        print(f'  lossblock_retina {name}: {time.time() - t0:.2f}s total='
              f'{float(total):.6f}', table.sum(1), 'pos', num_pos, 'ignored',
              int(sum((w == 0).sum() for w in label_weights)), 'vlr',
              int(sum((v > 0).sum() for v in vlr)))
    np.savez_compressed(os.path.join(OUT, 'lossblock_retina.npz'), **d)


def gen_e2e_retina():
    """One LD train step of configs/ld/ld_retina_r50_1x.py (LDRetinaHead
    student <- RetinaGFL R101 teacher) executed by the reference."""
    d = {}
    for name, pad, img_shape, num_gt, bseed in (
            ('tiny', (128, 160), (128, 150), [3, 2], 41),
            ('small', (256, 320), (256, 320), [5, 2], 42)):
        torch.manual_seed(0)
        det = build_reference_detector('configs/ld/ld_retina_r50_1x.py')
        det.load_state_dict(
            synthetic.seeded_state_dict(det.state_dict(), seed=1))
        det.teacher_model.load_state_dict(
            synthetic.seeded_state_dict(det.teacher_model.state_dict(),
                                        seed=2))
        det.train()
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        losses = det.forward_train(batch['img'], batch['img_metas'],
                                   batch['gt_bboxes'], batch['gt_labels'])
        table = np.stack([np.array([float(v.detach()) for v in losses[k]])
                          for k in RETINA_KEYS])
        loss, log_vars = det._parse_losses(losses)
        loss.backward()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        names, norms, proj = [], [], []
        for k, p in det.named_parameters():
            if p.grad is not None:
                names.append(k)
                norms.append(float(p.grad.double().norm()))
                gflat = p.grad.double().reshape(-1).numpy()
                proj.append([float(gflat @ synthetic.grad_probe(gflat.size,
                                                                sd))
                             for sd in (0, 1)])
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_norms'] = np.array(norms)
        d[name + '_grad_proj'] = np.array(proj)
        d[name + '_student_keys'] = np.array(list(det.state_dict().keys()))
        d[name + '_teacher_keys'] = np.array(
            list(det.teacher_model.state_dict().keys()))
        print(f'  e2e_retina {name}:',
              {k: round(v, 6) for k, v in log_vars.items()})
    np.savez_compressed(os.path.join(OUT, 'e2e_retina.npz'), **d)


def gen_e2e_atss(cfg_path='configs/ld/ld_r50_atss_r101_1x.py',
                 out_name='e2e_atss.npz', tag='e2e_atss'):
    """One LD train step of configs/ld/ld_r50_atss_r101_1x.py (LDATSSHead
    student <- ATSS-GFL R101 teacher) -- or, with the other arguments, of
    configs/ld/ld_r50_fcos_r101_1x.py -- executed by the reference: loss table,
    gradient norms and projections."""
    d = {}
    for name, pad, img_shape, num_gt, bseed in (
            ('tiny', (128, 160), (128, 150), [3, 2], 41),
            ('small', (256, 320), (256, 320), [5, 2], 42)):
        torch.manual_seed(0)
        det = build_reference_detector(cfg_path)
        det.load_state_dict(
            synthetic.seeded_state_dict(det.state_dict(), seed=1))
        det.teacher_model.load_state_dict(
            synthetic.seeded_state_dict(det.teacher_model.state_dict(),
                                        seed=2))
        det.train()
        batch = synthetic.synthetic_batch(
            num_imgs=len(num_gt), img_shape=img_shape, pad_shape=pad,
            num_gt=num_gt, seed=bseed)
        losses = det.forward_train(batch['img'], batch['img_metas'],
                                   batch['gt_bboxes'], batch['gt_labels'])
        table = np.stack([np.array([float(v.detach()) for v in losses[k]])
                          for k in ATSS_KEYS])
        loss, log_vars = det._parse_losses(losses)
        loss.backward()
        d[name + '_cfg'] = np.array(list(pad) + list(img_shape) + [bseed])
        d[name + '_num_gt'] = np.array(num_gt)
        d[name + '_losses'] = table.astype(np.float64)
        names, norms, proj = [], [], []
        for k, p in det.named_parameters():
            if p.grad is not None:
                names.append(k)
                norms.append(float(p.grad.double().norm()))
                gflat = p.grad.double().reshape(-1).numpy()
                proj.append([float(gflat @ synthetic.grad_probe(gflat.size,
                                                                sd))
                             for sd in (0, 1)])
        d[name + '_grad_names'] = np.array(names)
        d[name + '_grad_norms'] = np.array(norms)
        d[name + '_grad_proj'] = np.array(proj)
        d[name + '_student_keys'] = np.array(list(det.state_dict().keys()))
        d[name + '_teacher_keys'] = np.array(
            list(det.teacher_model.state_dict().keys()))
        print(f'  {tag} {name}:',
              {k: round(v, 6) for k, v in log_vars.items()})
    np.savez_compressed(os.path.join(OUT, out_name), **d)


def gen_infer_voting():
    """Score-voting Cluster-DIoU-NMS: the reference's own multiclass_nms branch
    (bbox_nms.py:141-176, pure torch -> PINNED, unlike the mmcv batched_nms of
    the default branch)."""
    import mmcv
    head = _ld_head()
    d = {}
    for case in synthetic.VOTING_CASES:
        name, pad, img_shapes, sfs, seed, nms_pre, clustered = case
        cls, reg, metas = synthetic.voting_inputs(case)
        for thr in (0.6, 0.85):
            cfg = mmcv.ConfigDict(dict(
                nms_pre=nms_pre, min_bbox_size=0, score_thr=0.05,
                nms=dict(type='voting_cluster_diounms', iou_threshold=thr),
                max_per_img=100))
            for rescale in (False, True):
                res = head.get_bboxes(cls, reg, metas, cfg=cfg,
                                      rescale=rescale)
                tag = f'{name}_t{int(thr * 100)}_r{int(rescale)}'
                for i, (db, dl) in enumerate(res):
                    d[f'{tag}_bboxes_{i}'] = _np(db).astype(np.float32)
                    d[f'{tag}_labels_{i}'] = _np(dl).astype(np.int64)
        # how much the voting moved the boxes (vs plain nms): fingerprint
        cfg0 = mmcv.ConfigDict(dict(nms_pre=nms_pre, min_bbox_size=0,
                                    score_thr=0.05,
                                    nms=dict(type='nms', iou_threshold=0.6),
                                    max_per_img=100))
        plain = head.get_bboxes(cls, reg, metas, cfg=cfg0, rescale=False)
        print(f'[voting] {name}: dets',
              [int(d[f"{name}_t60_r0_labels_{i}"].shape[0])
               for i in range(len(img_shapes))], 'plain nms dets',
              [int(p[1].shape[0]) for p in plain])
    np.savez_compressed(os.path.join(OUT, 'infer_voting.npz'), **d)


def gen_pipeline():
    """Input pipeline: the reference's OWN samplers,
    box transforms and random draws (pure numpy / torch code under
    /root/reference; the image arithmetic itself lives in mmcv / cv2, absent,
    see oracle/pipeline_oracle.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'ref_group_sampler',
        os.path.join(ref_shim.REFERENCE_ROOT,
                     'mmdet/datasets/samplers/group_sampler.py'))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    from mmdet.datasets.pipelines import transforms as T

    class DS:
        pass

    out = {}
    rs = np.random.RandomState(5)
    cases = {'mixed103': (rs.rand(103) < 0.7).astype(np.uint8),
             'one_group17': np.ones(17, np.uint8),
             'tiny3': np.array([0, 1, 1], np.uint8)}
    for name, flag in cases.items():
        ds = DS()
        ds.flag = flag
        out[f'sampler_{name}_flag'] = flag
        for spg in (1, 2):
            for world in (1, 2, 8):
                for seed, epoch in ((0, 0), (0, 3), (7, 1)):
                    rows = []
                    for rank in range(world):
                        s = gs.DistributedGroupSampler(ds, spg, world, rank,
                                                       seed=seed)
                        s.set_epoch(epoch)
                        rows.append(np.array(list(iter(s)), np.int64))
                        assert len(rows[-1]) == len(s)
                    out[f'sampler_{name}_spg{spg}_w{world}_s{seed}_e{epoch}'] = \
                        np.stack(rows)
        for spg in (1, 2, 4):
            np.random.seed(11 + spg)
            s = gs.GroupSampler(ds, spg)
            out[f'gsampler_{name}_spg{spg}'] = np.array(list(iter(s)), np.int64)
    # box transforms
    resize = T.Resize(img_scale=(1333, 800), keep_ratio=True)
    flipper = T.RandomFlip(flip_ratio=0.5)
    rs = np.random.RandomState(9)
    for i, (h, w, nh, nw) in enumerate(((480, 640, 800, 1067),
                                        (640, 427, 1199, 800),
                                        (375, 500, 800, 1067),
                                        (333, 1000, 444, 1333))):
        xy = rs.rand(23, 2) * [w, h]
        wh = rs.rand(23, 2) * [w, h] * 0.6
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        sf = np.array([nw / w, nh / h, nw / w, nh / h], np.float32)
        res = dict(gt_bboxes=boxes.copy(), bbox_fields=['gt_bboxes'],
                   scale_factor=sf, img_shape=(nh, nw, 3))
        resize._resize_bboxes(res)
        out[f'box{i}_in'] = boxes
        out[f'box{i}_geom'] = np.array([h, w, nh, nw])
        out[f'box{i}_resized'] = res['gt_bboxes']
        out[f'box{i}_flipped'] = flipper.bbox_flip(res['gt_bboxes'],
                                                   (nh, nw, 3), 'horizontal')
    # random draws: multi-scale modes and the flip decision
    scales = [(1333, 640), (1333, 800)]
    np.random.seed(21)
    out['draw_range'] = np.array(
        [T.Resize.random_sample(scales)[0] for _ in range(16)])
    np.random.seed(22)
    out['draw_value'] = np.array(
        [T.Resize.random_select([(1333, 640), (1333, 672), (1333, 800)])[0]
         for _ in range(16)])
    np.random.seed(23)
    flips = []
    for _ in range(32):
        res = dict(img=np.zeros((4, 4, 3), np.uint8), img_shape=(4, 4, 3),
                   img_fields=['img'], bbox_fields=[], mask_fields=[],
                   seg_fields=[])
        flips.append(bool(flipper(res)['flip']))
    out['draw_flip'] = np.array(flips)
    np.savez_compressed(os.path.join(OUT, 'pipeline.npz'), **out)
    print('pipeline.npz', len(out), 'arrays')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='kat,anchors,targets,lossblock,e2e,infer,'
                    'lossblock_v2,e2e_v2,imitation,pipeline,infer_voting,'
                    'lossblock_atss,e2e_atss,lossblock_fcos,e2e_fcos,'
                    'lossblock_retina,e2e_retina,infer_v2,infer_ctr,'
                    'infer_retina')
    ap.add_argument('--e2e-cases', default='')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    only = args.only.split(',')
    if 'kat' in only:
        gen_kat()
    if 'anchors' in only:
        gen_anchors()
    if 'targets' in only:
        gen_targets()
    if 'lossblock' in only:
        gen_lossblock()
    if 'e2e' in only:
        gen_e2e([c for c in args.e2e_cases.split(',') if c])
    if 'infer' in only:
        gen_infer()
    if 'imitation' in only:
        gen_imitation()
    if 'lossblock_v2' in only:
        gen_lossblock_v2()
    if 'e2e_v2' in only:
        gen_e2e_v2()
    if 'pipeline' in only:
        gen_pipeline()
    if 'infer_voting' in only:
        gen_infer_voting()
    if 'lossblock_atss' in only:
        gen_lossblock_atss()
    if 'e2e_atss' in only:
        gen_e2e_atss()
    if 'lossblock_fcos' in only:
        gen_lossblock_fcos()
    if 'e2e_fcos' in only:
        gen_e2e_atss('configs/ld/ld_r50_fcos_r101_1x.py', 'e2e_fcos.npz',
                     'e2e_fcos')
    if 'infer_v2' in only:
        gen_infer_v2()
    if 'infer_ctr' in only:
        gen_infer_ctr()
    if 'infer_retina' in only:
        gen_infer_retina()
    if 'lossblock_retina' in only:
        gen_lossblock_retina()
    if 'e2e_retina' in only:
        gen_e2e_retina()
    if 'resnext' in only:
        gen_resnext()
    if 'e2e_v2_r3' in only:
        gen_e2e_v2_r3([c for c in args.e2e_cases.split(',') if c])
    if 'lossblock_2rank' in only:
        gen_lossblock_2rank()
    if 'grad_truth64' in only:
        gen_grad_truth64()
    if 'grad_samples' in only:
        gen_grad_samples([c for c in args.e2e_cases.split(',') if c])


if __name__ == '__main__':
    main()
