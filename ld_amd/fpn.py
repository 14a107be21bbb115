"""FPN neck with mmdet's constructor and state_dict keys (reference:
mmdet/models/necks/fpn.py:11-221) on the HIP kernels."""
import torch.nn as nn

from . import layers as Y
from .cnn import Conv2d, ConvModule, xavier_init
from .registry import NECKS


@NECKS.register_module()
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0,
                 end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False,
                 no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        self.upsample_cfg = dict(upsample_cfg)
        if self.upsample_cfg != dict(mode='nearest'):
            raise NotImplementedError('only nearest top-down upsampling')
        if relu_before_extra_convs:
            raise NotImplementedError('relu_before_extra_convs')
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        self.add_extra_convs = add_extra_convs
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            self.add_extra_convs = 'on_input' if extra_convs_on_inputs \
                else 'on_output'

        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(
                ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                           norm_cfg=norm_cfg if not no_norm_on_lateral else
                           None, act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(
                ConvModule(out_channels, out_channels, 3, padding=1,
                           conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                           act_cfg=act_cfg, inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                if i == 0 and self.add_extra_convs == 'on_input':
                    ch = self.in_channels[self.backbone_end_level - 1]
                else:
                    ch = out_channels
                self.fpn_convs.append(
                    ConvModule(ch, out_channels, 3, stride=2, padding=1,
                               conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                               act_cfg=act_cfg, inplace=False))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, Conv2d):
                xavier_init(m, distribution='uniform')

    def forward(self, inputs):
        """fpn.py:170-221."""
        assert len(inputs) == len(self.in_channels)
        laterals = [
            lateral_conv(inputs[i + self.start_level])
            for i, lateral_conv in enumerate(self.lateral_convs)
        ]
        used = len(laterals)
        for i in range(used - 1, 0, -1):
            laterals[i - 1] = Y.upsample_add(laterals[i - 1], laterals[i])
        outs = [self.fpn_convs[i](laterals[i]) for i in range(used)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                raise NotImplementedError('max-pool extra levels')
            if self.add_extra_convs == 'on_input':
                src = inputs[self.backbone_end_level - 1]
            elif self.add_extra_convs == 'on_lateral':
                src = laterals[-1]
            else:
                src = outs[-1]
            outs.append(self.fpn_convs[used](src))
            for i in range(used + 1, self.num_outs):
                outs.append(self.fpn_convs[i](outs[-1]))
        return tuple(outs)
