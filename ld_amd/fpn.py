"""FPN neck with mmdet's constructor and state_dict keys (reference:
mmdet/models/necks/fpn.py:11-221) on the HIP kernels."""
import torch.nn as nn

import torch

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
        """Constructor arguments and module names (``lateral_convs.i.conv``,
        ``fpn_convs.i.conv``) of mmdet/models/necks/fpn.py:66-160."""
        super().__init__()
        if not isinstance(in_channels, list):
            raise AssertionError('in_channels must be a list')
        if dict(upsample_cfg) != dict(mode='nearest'):
            raise NotImplementedError('only nearest top-down upsampling')
        if not isinstance(add_extra_convs, (str, bool)):
            raise AssertionError('add_extra_convs: str or bool')
        n_in = len(in_channels)
        last = n_in if end_level == -1 else end_level
        used = last - start_level  # backbone levels that get a lateral conv
        if end_level == -1:
            assert num_outs >= used
        else:
            assert end_level <= n_in and num_outs == used
        # where the extra (stride-2) levels take their input from
        source = add_extra_convs
        if isinstance(source, str):
            assert source in ('on_input', 'on_lateral', 'on_output')
        elif source:
            source = 'on_input' if extra_convs_on_inputs else 'on_output'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = n_in, num_outs
        self.start_level, self.end_level = start_level, end_level
        self.backbone_end_level = last
        self.add_extra_convs = source
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        self.upsample_cfg = dict(upsample_cfg)

        def block(cin, k, **kw):
            return ConvModule(cin, out_channels, k, conv_cfg=conv_cfg,
                              act_cfg=act_cfg, inplace=False, **kw)

        lateral_norm = None if no_norm_on_lateral else norm_cfg
        self.lateral_convs = nn.ModuleList(
            block(in_channels[i], 1, norm_cfg=lateral_norm)
            for i in range(start_level, last))
        self.fpn_convs = nn.ModuleList(
            block(out_channels, 3, padding=1, norm_cfg=norm_cfg)
            for _ in range(used))
        if source:
            for j in range(num_outs - used):
                cin = in_channels[last - 1] \
                    if (j == 0 and source == 'on_input') else out_channels
                self.fpn_convs.append(block(cin, 3, stride=2, padding=1,
                                            norm_cfg=norm_cfg))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, Conv2d):
                xavier_init(m, distribution='uniform')

    def _relu(self, x):
        """F.relu as the library's affine + ReLU launch with the identity
        affine (scale = 1 / sqrt(1 + 0) = 1 and shift = 0 exactly)."""
        c = x.shape[1]
        if getattr(self, '_relu_c', None) is None or \
                self._relu_c[0].device != x.device:
            one = torch.ones(c, device=x.device)
            zero = torch.zeros(c, device=x.device)
            self._relu_c = (one, zero)
        one, zero = self._relu_c
        n, _, h, w = x.shape
        y = Y.bn_act(x.reshape(n, c, h * w), one, zero, zero, one, 0.0,
                     relu=True)
        return y.reshape(n, c, h, w)

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
                prev = outs[-1]
                if self.relu_before_extra_convs:  # fpn.py:213-216
                    prev = self._relu(prev)
                outs.append(self.fpn_convs[i](prev))
        return tuple(outs)
