"""Model config dicts for the LD benchmark configurations, built in code.

The reference's own config files (configs/ld/*.py, configs/gfl/*.py) resolve
through ld_amd's registry unchanged (tests/test_configs_resolve.py checks this
wherever a reference checkout is present); they are not shipped with this
repository, so the bench / smoke / GPU tests construct the same settings here.
Values follow configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py:6-58,
configs/ld/ld_r50_gflv1_r101_fpn_coco_1x.py:2-52 and
configs/gfl/gfl_r50_fpn_1x_coco.py:5-56.
"""
import copy

_RESNET_CH = {18: [64, 128, 256, 512], 34: [64, 128, 256, 512],
              50: [256, 512, 1024, 2048], 101: [256, 512, 1024, 2048]}

_TRAIN_CFG = dict(assigner=dict(type='ATSSAssigner', topk=9),
                  allowed_border=-1, pos_weight=-1, debug=False)
_TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05,
                 nms=dict(type='nms', iou_threshold=0.6), max_per_img=100)


def _backbone(depth):
    return dict(type='ResNet', depth=depth, num_stages=4,
                out_indices=(0, 1, 2, 3), frozen_stages=1,
                norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                style='pytorch')


def _x101_backbone(depth=101, dcn=False):
    """The ResNeXt-101 32x4d teacher backbone of
    configs/imv2/gflv2_x101_fpn_2x_coco.py:8-20.  ``dcn=True`` keeps that
    file's grouped DCN in c4-c5 (parity unpinned: mmcv's op is absent from the
    reference checkout); the pinned composition runs without it."""
    cfg = dict(type='ResNeXt', depth=depth, groups=32, base_width=4,
               num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
               norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
               style='pytorch')
    if dcn:
        cfg.update(dcn=dict(type='DCN', deform_groups=1,
                            fallback_on_stride=False),
                   stage_with_dcn=(False, False, True, True))
    return cfg


def _neck(depth):
    return dict(type='FPN', in_channels=list(_RESNET_CH[depth]),
                out_channels=256, start_level=1, add_extra_convs='on_output',
                num_outs=5)


def _gfl_head_common():
    return dict(
        num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0],
                              octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0,
                      loss_weight=1.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
        reg_max=16)


def gfl_detector(depth=101):
    """A GFL teacher (configs/gfl/gfl_r{50,101}_fpn_*_coco.py)."""
    head = dict(type='GFLHead', loss_bbox=dict(type='CIoULoss',
                                               loss_weight=2.0),
                **_gfl_head_common())
    return dict(type='GFL', pretrained=None, backbone=_backbone(depth),
                neck=_neck(depth), bbox_head=head,
                train_cfg=copy.deepcopy(_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def ld_detector(student_depth=50, teacher_depth=101,
                imitation_method='finegrained', loss_im_weight=2.0,
                with_vlr_kd=True):
    """KnowledgeDistillationSingleStageDetector as in
    configs/ld/ld_r50_gflv1_r101_fpn_coco_1x.py (Main KD + Main LD + VLR LD +
    fine-grained feature imitation).  ``with_vlr_kd=False`` gives the r18/r34
    style config that only sets loss_ld."""
    head = dict(type='LDHead', loss_bbox=dict(type='GIoULoss',
                                              loss_weight=2.0),
                loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=0.25, T=10),
                **_gfl_head_common())
    if with_vlr_kd:
        head.update(
            loss_ld_vlr=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=0.25, T=10),
            loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                         loss_weight=10, T=2))
    head.update(loss_im=dict(type='IMLoss', loss_weight=loss_im_weight),
                imitation_method=imitation_method)
    return dict(type='KnowledgeDistillationSingleStageDetector',
                pretrained=None,
                teacher_config=dict(model=gfl_detector(teacher_depth)),
                teacher_ckpt=None, output_feature=True,
                backbone=_backbone(student_depth), neck=_neck(student_depth),
                bbox_head=head, train_cfg=copy.deepcopy(_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def _gflv2_head_common():
    """configs/gfl/gflv2_r50_fpn_1x_coco.py:20-40: GFLv2's QFL runs on the
    joint probability (use_sigmoid=False)."""
    h = _gfl_head_common()
    h['loss_cls'] = dict(type='QualityFocalLoss', use_sigmoid=False, beta=2.0,
                         loss_weight=1.0)
    h.update(reg_topk=4, reg_channels=64, add_mean=True)
    return h


def gflv2_detector(depth=101):
    """A GFLv2 teacher (configs/gfl/gflv2_r101_fpn_2x_coco.py)."""
    head = dict(type='GFocalHead', loss_bbox=dict(type='GIoULoss',
                                                  loss_weight=2.0),
                **_gflv2_head_common())
    return dict(type='GFL', pretrained=None, backbone=_backbone(depth),
                neck=_neck(depth), bbox_head=head,
                train_cfg=copy.deepcopy(_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def ldv2_x101_detector(student_depth=50, imitation_method='finegrained',
                       loss_im_weight=2.0, dcn=False):
    """BASELINE config 5 as worded -- "ldv2 + imitation_method='finegrain',
    R50 <- X101": configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py whose GFLv2 teacher
    takes the ResNeXt-101 32x4d backbone of
    configs/imv2/gflv2_x101_fpn_2x_coco.py:8-20 (the reference
    ships no such file; this is the composition)."""
    cfg = ldv2_detector(student_depth, 101, imitation_method, loss_im_weight)
    cfg['teacher_config']['model']['backbone'] = _x101_backbone(101, dcn)
    return cfg


def ldv2_detector(student_depth=50, teacher_depth=101,
                  imitation_method='finegrained', loss_im_weight=2.0):
    """configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py with the device-clean
    imitation method (its own 'gibox' is CUDA-only in the reference, SURVEY
    quirk Q3)."""
    head = dict(type='LDv2Head', loss_bbox=dict(type='GIoULoss',
                                                loss_weight=2.0),
                loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=0.25, T=10),
                loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=10, T=2),
                loss_im=dict(type='IMLoss', loss_weight=loss_im_weight),
                imitation_method=imitation_method, **_gflv2_head_common())
    return dict(type='KnowledgeDistillationSingleStageDetector',
                pretrained=None,
                teacher_config=dict(model=gflv2_detector(teacher_depth)),
                teacher_ckpt=None, output_feature=True,
                backbone=_backbone(student_depth), neck=_neck(student_depth),
                bbox_head=head, train_cfg=copy.deepcopy(_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def _atss_head_common():
    """configs/gfl/atss_gfl_r50_1x.py:26-47 / configs/ld/ld_r50_atss_r101_1x.py:
    29-58."""
    return dict(
        num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0],
                              octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                        target_means=[.0, .0, .0, .0],
                        target_stds=[0.1, 0.1, 0.2, 0.2]),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True,
                             loss_weight=1.0))


def atss_gfl_detector(depth=101):
    """An ATSS-GFL teacher (configs/gfl/atss_gfl_r{50,101}_*.py)."""
    return dict(type='ATSS', pretrained=None, backbone=_backbone(depth),
                neck=_neck(depth),
                bbox_head=dict(type='ATSSGFLHead', **_atss_head_common()),
                train_cfg=copy.deepcopy(_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def ld_atss_detector(student_depth=50, teacher_depth=101):
    """configs/ld/ld_r50_atss_r101_1x.py: LDATSSHead student <- ATSS-GFL
    teacher; output_feature is the detector's default (False): the head takes
    no teacher features."""
    head = dict(type='LDATSSHead',
                loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=0.25, T=10),
                loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=10, T=2),
                **_atss_head_common())
    return dict(type='KnowledgeDistillationSingleStageDetector',
                pretrained=None,
                teacher_config=dict(model=atss_gfl_detector(teacher_depth)),
                teacher_ckpt=None, backbone=_backbone(student_depth),
                neck=_neck(student_depth), bbox_head=head,
                train_cfg=copy.deepcopy(_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def _fcos_backbone(depth):
    """configs/gfl/fcos_gfl_r50_center.py:4-12: caffe-style ResNet, BN frozen."""
    return dict(type='ResNet', depth=depth, num_stages=4,
                out_indices=(0, 1, 2, 3), frozen_stages=1,
                norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True,
                style='caffe')


def _fcos_neck(depth):
    return dict(type='FPN', in_channels=list(_RESNET_CH[depth]),
                out_channels=256, start_level=1, add_extra_convs=True,
                extra_convs_on_inputs=False, num_outs=5,
                relu_before_extra_convs=True)


def _fcos_head_common():
    """configs/gfl/fcos_gfl_r50_center.py:22-43."""
    return dict(
        num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
        strides=[8, 16, 32, 64, 128],
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=1.0),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True,
                             loss_weight=1.0),
        norm_on_bbox=False, centerness_on_reg=True, dcn_on_last_conv=False,
        center_sampling=True, conv_bias=True)


_FCOS_TRAIN_CFG = dict(
    assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4,
                  min_pos_iou=0, ignore_iof_thr=-1),
    allowed_border=-1, pos_weight=-1, debug=False)


def fcos_gfl_detector(depth=101):
    """An FCOS-GFL teacher (configs/gfl/fcos_gfl_r{50,101}_*center.py)."""
    return dict(type='FCOS', pretrained=None, backbone=_fcos_backbone(depth),
                neck=_fcos_neck(depth),
                bbox_head=dict(type='FCOSGFLHead', **_fcos_head_common()),
                train_cfg=copy.deepcopy(_FCOS_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def ld_fcos_detector(student_depth=50, teacher_depth=101):
    """configs/ld/ld_r50_fcos_r101_1x.py: LDFCOSHead student <- FCOS-GFL
    teacher (output_feature=False)."""
    head = dict(type='LDFCOSHead',
                loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=0.25, T=10),
                loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=10, T=2),
                **_fcos_head_common())
    return dict(type='KnowledgeDistillationSingleStageDetector',
                pretrained=None,
                teacher_config=dict(model=fcos_gfl_detector(teacher_depth)),
                teacher_ckpt=None, backbone=_fcos_backbone(student_depth),
                neck=_fcos_neck(student_depth), bbox_head=head,
                train_cfg=copy.deepcopy(_FCOS_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


OPTIMIZER = dict(type='SGD', lr=0.0025, momentum=0.9, weight_decay=0.0001)


def ld_r101_dcn_detector(loss_im_weight=0.0):
    """BASELINE config 4, configs/ld/ld_r101_gflv1_r101dcn_fpn_coco_2x.py: the
    r18 config's head (loss_ld only; its default 'gibox' imitation has weight 0,
    evaluated as 'finegrained' x 0 like config 1) on a ResNet-101
    student, with the R101 teacher of
    configs/gfl/gfl_r101_fpn_dconv_c3-c5_mstrain_2x_coco.py:6-12 (DCNv1 in
    c3-c5)."""
    cfg = ld_detector(101, 101, loss_im_weight=loss_im_weight,
                      with_vlr_kd=False)
    cfg['teacher_config']['model']['backbone'].update(
        dcn=dict(type='DCN', deform_groups=1, fallback_on_stride=False),
        stage_with_dcn=(False, True, True, True))
    return cfg


def build_seeded(cfg, device=None, student_seed=1, teacher_seed=2):
    """Any KD detector config with the deterministic synthetic weights."""
    from . import synthetic
    from .registry import build_detector
    det = build_detector(cfg)
    det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(),
                                                    seed=student_seed))
    det.teacher_model.load_state_dict(
        synthetic.seeded_state_dict(det.teacher_model.state_dict(),
                                    seed=teacher_seed))
    if device is not None:
        det.to(device)
    det.train()
    return det


def build_seeded_ld_detector(student_depth=50, teacher_depth=101, device=None,
                             loss_im_weight=2.0, student_seed=1,
                             teacher_seed=2):
    """Detector with the deterministic synthetic weights the golden fixtures
    were generated with (ld_amd.synthetic.seeded_state_dict)."""
    from . import synthetic
    from .registry import build_detector
    det = build_detector(ld_detector(student_depth, teacher_depth,
                                     loss_im_weight=loss_im_weight))
    det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(),
                                                    seed=student_seed))
    det.teacher_model.load_state_dict(
        synthetic.seeded_state_dict(det.teacher_model.state_dict(),
                                    seed=teacher_seed))
    if device is not None:
        det.to(device)
    det.train()
    return det


# ---------------------------------------------------------------- RetinaGFL --
def _retina_head_common():
    return dict(
        num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4,
                              scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                              strides=[8, 16, 32, 64, 128]),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder',
                        target_means=[.0, .0, .0, .0],
                        target_stds=[1.0, 1.0, 1.0, 1.0]),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0,
                      alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        reg_decoded_bbox=True)


def _retina_neck(depth=50):
    return dict(type='FPN', in_channels=list(_RESNET_CH[depth]),
                out_channels=256, start_level=1, add_extra_convs='on_input',
                num_outs=5)


def retina_gfl_detector(depth=101):
    """A RetinaGFL teacher (configs/gfl/retinagfl_r101_2x_coco.py)."""
    return dict(type='RetinaNet', pretrained=None,
                backbone=_backbone(depth), neck=_retina_neck(depth),
                bbox_head=dict(type='RetinaGFLHead', **_retina_head_common()),
                train_cfg=copy.deepcopy(_FCOS_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))


def ld_retina_detector(student_depth=50, teacher_depth=101):
    """configs/ld/ld_retina_r50_1x.py: LDRetinaHead student <- RetinaGFL
    teacher (output_feature=False)."""
    head = dict(type='LDRetinaHead',
                loss_ld=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=5, T=10),
                loss_kd=dict(type='KnowledgeDistillationKLDivLoss',
                             loss_weight=10, T=8),
                **_retina_head_common())
    return dict(type='KnowledgeDistillationSingleStageDetector',
                pretrained=None,
                teacher_config=dict(model=retina_gfl_detector(teacher_depth)),
                teacher_ckpt=None, output_feature=False,
                backbone=_backbone(student_depth),
                neck=_retina_neck(student_depth),
                bbox_head=head, train_cfg=copy.deepcopy(_FCOS_TRAIN_CFG),
                test_cfg=copy.deepcopy(_TEST_CFG))
