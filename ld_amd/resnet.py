"""ResNet-18/34/50/101 backbone with mmdet's constructor, state_dict keys and
freezing semantics (reference: mmdet/models/backbones/resnet.py:13-648,
mmdet/models/utils/res_layer.py:5-102), running on the HIP kernels.

Execution plan per block:
  * no gradient needed (teacher under no_grad, frozen stem / stages whose
    input carries no grad): conv -> BN(eval) -> (+identity) -> ReLU is ONE
    implicit-GEMM launch with the BN folded into its epilogue;
  * trainable: ONE forward launch as well (layers.ConvBnActFn: the conv
    epilogue applies the folded eval-BN, the residual and the ReLU and also
    stores the raw conv result the gamma/beta gradients need); backward = the
    BN-affine/ReLU backward kernel + MFMA dgrad / wgrad.
"""
import math

import torch
import torch.nn as nn

from . import layers as Y
from . import lib as L
from .cnn import (BatchNorm2d, Conv2d, GroupedConv2d, build_conv_layer,
                  build_norm_layer, constant_init, kaiming_init)
from .registry import BACKBONES


def _conv_bn(x3, levels, conv, bn, residual=None, relu=True):
    """conv -> bn -> (+residual) -> relu with the right fusion level."""
    w = conv.weight
    need_grad = torch.is_grad_enabled() and (
        x3.requires_grad or w.requires_grad or bn.weight.requires_grad or
        (residual is not None and residual.requires_grad))
    if bn.training:
        raise NotImplementedError('BatchNorm in training mode (use '
                                  'norm_eval=True; resnet.py:639-648)')
    if hasattr(conv, 'forward3_fused'):  # DCN: forward-only, fused epilogue
        scale, shift, _ = Y.bn_prepare(bn.weight, bn.bias, bn.running_mean,
                                       bn.running_var, bn.eps)
        return conv.forward3_fused(x3, levels, scale, shift, residual, relu)
    if not need_grad:
        return Y.conv_bn_act_infer(x3, w, bn.weight, bn.bias, bn.running_mean,
                                   bn.running_var, bn.eps, conv.stride[0],
                                   conv.padding[0], levels, residual, relu)
    if type(conv) is Conv2d and conv.bias is None:
        # trainable pair: ONE forward launch (conv with the folded BN, residual
        # and ReLU in its epilogue + the raw conv result for the BN backward)
        return Y.conv_bn_act(x3, w, bn.weight, bn.bias, bn.running_mean,
                             bn.running_var, bn.eps, conv.stride[0],
                             conv.padding[0], levels, residual, relu)
    y3, lv = conv.forward3(x3, levels)
    return bn.forward3(y3, residual, relu), lv


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None,
                 style='pytorch', with_cp=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), dcn=None, plugins=None):
        super().__init__()
        assert dcn is None and plugins is None and dilation == 1
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3,
                                      stride=stride, padding=1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1,
                                      bias=False)
        self.add_module(self.norm2_name, norm2)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward3(self, x3, levels):
        out, lv = _conv_bn(x3, levels, self.conv1, self.norm1)
        identity = x3
        if self.downsample is not None:
            identity, _ = _conv_bn(x3, levels, self.downsample[0],
                                   self.downsample[1], relu=False)
        return _conv_bn(out, lv, self.conv2, self.norm2, residual=identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None,
                 style='pytorch', with_cp=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), dcn=None, plugins=None):
        super().__init__()
        assert style in ['pytorch', 'caffe']
        assert dcn is None or isinstance(dcn, dict)
        assert plugins is None and dilation == 1
        self.dcn, self.with_dcn = dcn, dcn is not None
        self.conv1_stride, self.conv2_stride = (1, stride) \
            if style == 'pytorch' else (stride, 1)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(
            norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 1,
                                      stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        fallback_on_stride = False
        if self.with_dcn:  # resnet.py:171-194
            dcn = dict(dcn)
            fallback_on_stride = dcn.pop('fallback_on_stride', False)
        if not self.with_dcn or fallback_on_stride:
            self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3,
                                          stride=self.conv2_stride, padding=1,
                                          bias=False)
        else:
            assert conv_cfg is None, 'conv_cfg must be None for DCN'
            self.conv2 = build_conv_layer(dcn, planes, planes, 3,
                                          stride=self.conv2_stride, padding=1,
                                          bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, planes,
                                      planes * self.expansion, 1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    @property
    def norm3(self):
        return getattr(self, self.norm3_name)

    def _fusable(self):
        """An identity block of plain convs with eval-mode norms: what
        layers.bottleneck_c8_forward runs as one launch when the trunk is frozen
        and C8-only (the R101 teacher's layer3, bf16 mode)."""
        ok = getattr(self, '_ld_fusable', None)
        if ok is None:
            ok = self.downsample is None and not self.with_dcn and \
                all(type(c) is Conv2d and c.bias is None and c.stride[0] == 1
                    for c in (self.conv1, self.conv2, self.conv3)) and \
                self.conv2.padding[0] == 1 and type(self) is Bottleneck
            self._ld_fusable = ok
        return ok and not any(n.training for n in
                              (self.norm1, self.norm2, self.norm3))

    def forward3(self, x3, levels):
        if isinstance(x3, Y.C8Act) and not torch.is_grad_enabled() and \
                self._fusable() and Y.fused_bottleneck_available(
                    x3, self.conv1.weight.shape[1], self.conv1.weight.shape[0],
                    levels):
            return Y.bottleneck_c8_forward(
                x3, levels,
                [c.weight for c in (self.conv1, self.conv2, self.conv3)],
                [(n.weight, n.bias, n.running_mean, n.running_var, n.eps)
                 for n in (self.norm1, self.norm2, self.norm3)])
        out, lv = _conv_bn(x3, levels, self.conv1, self.norm1)
        out, lv = _conv_bn(out, lv, self.conv2, self.norm2)
        identity = x3
        if self.downsample is not None:
            identity, _ = _conv_bn(x3, levels, self.downsample[0],
                                   self.downsample[1], relu=False)
        return _conv_bn(out, lv, self.conv3, self.norm3, residual=identity)


class ResLayer(nn.Sequential):
    """mmdet/models/utils/res_layer.py:5-102 (downsample_first=True,
    avg_down=False)."""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1,
                 avg_down=False, conv_cfg=None, norm_cfg=dict(type='BN'),
                 downsample_first=True, **kwargs):
        assert not avg_down and downsample_first
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                build_conv_layer(conv_cfg, inplanes, planes * block.expansion,
                                 1, stride=stride, bias=False),
                build_norm_layer(norm_cfg, planes * block.expansion)[1])
        layers = [block(inplanes=inplanes, planes=planes, stride=stride,
                        downsample=downsample, conv_cfg=conv_cfg,
                        norm_cfg=norm_cfg, **kwargs)]
        inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(inplanes=inplanes, planes=planes, stride=1,
                                conv_cfg=conv_cfg, norm_cfg=norm_cfg, **kwargs))
        super().__init__(*layers)


class ResNeXtBottleneck(Bottleneck):
    """Bottleneck block of ResNeXt (mmdet/models/backbones/resnext.py:11-85):
    conv1 / conv3 go through ``width = floor(planes * base_width /
    base_channels) * groups`` channels and conv2 is the grouped 3x3 (or a
    grouped DCN when the stage has ``dcn``).  Same module and state_dict names
    as the parent; the constructor builds the parent first and replaces the
    three convs / norms exactly as the reference does."""
    expansion = 4

    def __init__(self, inplanes, planes, groups=1, base_width=4,
                 base_channels=64, **kwargs):
        super().__init__(inplanes, planes, **kwargs)
        norm_cfg = kwargs.get('norm_cfg', dict(type='BN'))
        conv_cfg = kwargs.get('conv_cfg', None)
        width = planes if groups == 1 else \
            math.floor(planes * (base_width / base_channels)) * groups
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, width, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, width, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(
            norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, width, 1,
                                      stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        fallback_on_stride = False
        dcn = None
        if self.with_dcn:
            dcn = dict(self.dcn)
            fallback_on_stride = dcn.pop('fallback_on_stride', False)
        if not self.with_dcn or fallback_on_stride:
            self.conv2 = build_conv_layer(conv_cfg, width, width, 3,
                                          stride=self.conv2_stride, padding=1,
                                          groups=groups, bias=False)
        else:
            assert conv_cfg is None, 'conv_cfg must be None for DCN'
            self.conv2 = build_conv_layer(dcn, width, width, 3,
                                          stride=self.conv2_stride, padding=1,
                                          groups=groups, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, width,
                                      planes * self.expansion, 1, bias=False)
        self.add_module(self.norm3_name, norm3)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {
        18: (BasicBlock, (2, 2, 2, 2)),
        34: (BasicBlock, (3, 4, 6, 3)),
        50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)),
        152: (Bottleneck, (3, 8, 36, 3))
    }

    def __init__(self, depth, in_channels=3, stem_channels=64, base_channels=64,
                 num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False,
                 avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                 dcn=None, stage_with_dcn=(False, False, False, False),
                 plugins=None, with_cp=False, zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if deep_stem or avg_down or plugins is not None or \
                tuple(dilations) != (1, 1, 1, 1):
            raise NotImplementedError('ResNet variant outside the LD configs')
        self.dcn, self.stage_with_dcn = dcn, stage_with_dcn
        if dcn is not None:
            assert len(stage_with_dcn) == num_stages
            if self.arch_settings[depth][0] is BasicBlock:
                raise NotImplementedError('DCN in BasicBlock ResNets')
        self.depth, self.stem_channels = depth, stem_channels
        self.base_channels, self.num_stages = base_channels, num_stages
        assert 1 <= num_stages <= 4
        self.strides, self.out_indices = strides, out_indices
        assert max(out_indices) < num_stages
        self.style, self.frozen_stages = style, frozen_stages
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.norm_eval = norm_eval
        self.zero_init_residual = zero_init_residual
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = stem_channels

        self.conv1 = build_conv_layer(conv_cfg, in_channels, stem_channels, 7,
                                      stride=2, padding=3, bias=False)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, stem_channels,
                                                  postfix=1)
        self.add_module(self.norm1_name, norm1)

        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2**i
            stage_dcn = dcn if (dcn is not None and stage_with_dcn[i]) \
                else None
            res_layer = self.make_res_layer(
                block=self.block, inplanes=self.inplanes, planes=planes,
                num_blocks=num_blocks, stride=strides[i], style=self.style,
                conv_cfg=conv_cfg, norm_cfg=norm_cfg, dcn=stage_dcn)
            self.inplanes = planes * self.block.expansion
            layer_name = f'layer{i + 1}'
            self.add_module(layer_name, res_layer)
            self.res_layers.append(layer_name)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * base_channels * 2**(
            len(self.stage_blocks) - 1)

    def make_res_layer(self, **kwargs):
        """resnet.py:515-517 (the hook ResNeXt overrides)."""
        return ResLayer(**kwargs)

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    def _freeze_stages(self):
        """resnet.py:572-588."""
        if self.frozen_stages >= 0:
            self.norm1.eval()
            for m in [self.conv1, self.norm1]:
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        """resnet.py:590-620 (pretrained checkpoints need the checkpoint wire
        format, SURVEY.md section 8f-2: not available offline)."""
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, (Conv2d, GroupedConv2d)):
                    kaiming_init(m)
                elif isinstance(m, BatchNorm2d):
                    constant_init(m, 1)
            if self.dcn is not None:  # resnet.py:607-611
                for m in self.modules():
                    if isinstance(m, Bottleneck) and \
                            hasattr(m.conv2, 'conv_offset'):
                        constant_init(m.conv2.conv_offset, 0)
            if self.zero_init_residual:
                for m in self.modules():
                    if isinstance(m, Bottleneck):
                        constant_init(m.norm3, 0)
                    elif isinstance(m, BasicBlock):
                        constant_init(m.norm2, 0)
        else:
            raise TypeError('pretrained must be a str or None')

    def forward(self, x):
        """(N, 3, H, W) -> tuple of stage outputs (resnet.py:622-637)."""
        L.require_device(x, torch.float32, 'backbone input')
        n, c, h, w = x.shape
        if self.frozen_stages < 0 and torch.is_grad_enabled():
            raise NotImplementedError(
                'a trainable stem needs a max-pool backward; the LD configs '
                'freeze it (frozen_stages=1)')
        x3 = x.reshape(n, c, h * w)
        x3, lv = _conv_bn(x3, ((h, w), ), self.conv1, self.norm1)
        x4 = Y.maxpool3x3s2(x3.view(n, -1, lv[0][0], lv[0][1]))
        lv = ((x4.shape[2], x4.shape[3]), )
        x3 = x4.reshape(n, x4.shape[1], -1)
        if self._c8_only():
            # frozen teacher, bf16 mode: the trunk lives only as bf16 C8 images
            with Y.c8_only_scope():
                return self._stages(Y.C8Act(Y.to_c8(x3), x3.shape), lv, n)
        return self._stages(x3, lv, n, self._frozen_c8_stages())

    def _frozen_c8_stages(self):
        """bf16 mode, training student: its FROZEN leading stages (frozen_stages
        = 1 in every LD config: stem + layer1, resnet.py:572-588) are read by
        convs only and take no gradient, exactly like the teacher's trunk -- they
        run C8-only too (VERDICT round 2, next #1a: the 200x336 stage's fp32
        outputs were 2/3 of the bytes of its HBM-bound 1x1 convs).  The first
        trainable block then takes the C8 image as its conv operand, forward and
        weight gradient (layers.ConvFn).  Returns how many leading stages."""
        if self.frozen_stages < 1 or not torch.is_grad_enabled() or \
                getattr(self, 'c8_activations', False) or \
                not Y.WGRAD_C8_ON():
            return 0
        k = min(self.frozen_stages, len(self.res_layers) - 1)
        cached = getattr(self, '_c8_frozen', None)
        if cached is None or cached[0] != k:
            convs = [m for name in self.res_layers[:k + 1]
                     for m in getattr(self, name).modules()
                     if hasattr(m, 'weight') and m.weight.dim() == 4]
            plain = not any(hasattr(m, 'forward3_fused') for m in convs)
            frozen = all(not p.requires_grad for name in self.res_layers[:k]
                         for p in getattr(self, name).parameters())
            # the first trainable stage must start with a downsample block:
            # its identity path is then a conv of the C8 image, never the
            # image itself
            nxt = getattr(self, self.res_layers[k])[0]
            chans = sorted({c for m in convs for c in m.weight.shape[:2]})
            self._c8_frozen = (k, plain and frozen and
                               nxt.downsample is not None, chans)
            cached = self._c8_frozen
        return k if cached[1] and Y.c8_only_available(cached[2]) else 0

    def _stages(self, x3, lv, n, c8_stages=0):
        outs = []
        if c8_stages:
            x3 = Y.C8Act(Y.to_c8(x3), x3.shape)
        for i, layer_name in enumerate(self.res_layers):
            if i < c8_stages:
                with Y.c8_only_scope():
                    for blk in getattr(self, layer_name):
                        x3, lv = blk.forward3(x3, lv)
            else:
                for blk in getattr(self, layer_name):
                    x3, lv = blk.forward3(x3, lv)
            if i in self.out_indices:
                outs.append(x3.view(n, x3.shape[1], lv[0][0], lv[0][1]))
        return tuple(outs)

    def _c8_only(self):
        """``c8_activations`` (set by the KD detector on its frozen teacher):
        under no_grad, with every BN in eval mode and no DCN block, the stages
        keep only bf16 C8 activations (layers.C8Act) and the neck's lateral
        convs read those."""
        if not getattr(self, 'c8_activations', False) or \
                torch.is_grad_enabled() or self.training:
            return False
        plain = getattr(self, '_c8_plain', None)
        if plain is None:
            convs = [m for m in self.modules()
                     if hasattr(m, 'weight') and m.weight.dim() == 4]
            plain = not any(hasattr(m, 'forward3_fused') for m in convs)
            stage = [m for name in self.res_layers
                     for m in getattr(self, name).modules()
                     if hasattr(m, 'weight') and m.weight.dim() == 4]
            self._c8_channels = sorted({c for m in stage
                                        for c in m.weight.shape[:2]})
            self._c8_plain = plain
        return plain and Y.c8_only_available(self._c8_channels)

    def train(self, mode=True):
        """resnet.py:639-648: keep frozen stages and (norm_eval) all BN in
        eval mode."""
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, BatchNorm2d):
                    m.eval()
        return self


@BACKBONES.register_module()
class ResNeXt(ResNet):
    """ResNeXt backbone (mmdet/models/backbones/resnext.py:88-153): ResNet with
    grouped Bottlenecks.  Here it is the frozen X-101 (32x4d) TEACHER of
    BASELINE config 5 -- its grouped convs are forward-only
    (ld_amd/csrc/gconv.hip), so it must run under ``torch.no_grad()`` with
    ``norm_eval=True``."""
    arch_settings = {
        50: (ResNeXtBottleneck, (3, 4, 6, 3)),
        101: (ResNeXtBottleneck, (3, 4, 23, 3)),
        152: (ResNeXtBottleneck, (3, 8, 36, 3))
    }

    def __init__(self, groups=1, base_width=4, **kwargs):
        self.groups, self.base_width = groups, base_width
        super().__init__(**kwargs)

    def make_res_layer(self, **kwargs):
        return ResLayer(groups=self.groups, base_width=self.base_width,
                        base_channels=self.base_channels, **kwargs)
