"""Building blocks with mmcv.cnn's names and state_dict keys (ConvModule,
Scale, build_conv_layer, build_norm_layer, init helpers), implemented over the
HIP layer functions in ld_amd.layers.  Parameter shapes and key names match
mmdet/mmcv so released checkpoints map 1:1 (SURVEY.md section 5 "checkpoint").

Modules take and return ordinary (N, C, H, W) tensors; the level-concatenated
fast path of the head lives in ld_amd.heads.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import layers as Y


# ------------------------------------------------------------------ inits --
def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if getattr(module, 'weight', None) is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0,
                 distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if getattr(module, 'weight', None) is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode,
                                     nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode,
                                    nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


# ---------------------------------------------------------------- layers ---
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2d(nn.Module):
    """nn.Conv2d's parameters and keys; forward = MFMA implicit GEMM."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        k, s, p, d = (_pair(kernel_size), _pair(stride), _pair(padding),
                      _pair(dilation))
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1]:
            raise NotImplementedError('only square kernels/strides/pads')
        if d != (1, 1) or groups != 1:
            raise NotImplementedError(
                'dilated / grouped convs are not on the LD hot path')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = k, s, p
        self.weight = nn.Parameter(
            torch.empty(out_channels, in_channels, k[0], k[1]))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward3(self, x3, levels):
        """(N, C, P) level-concatenated in/out."""
        return Y.conv2d(x3, self.weight, self.bias, self.stride[0],
                        self.padding[0], levels)

    def forward(self, x):
        n, c, h, w = x.shape
        y3, lv = self.forward3(x.reshape(n, c, h * w), ((h, w), ))
        return y3.view(n, self.out_channels, lv[0][0], lv[0][1])

    def extra_repr(self):
        return (f'{self.in_channels}, {self.out_channels}, '
                f'kernel_size={self.kernel_size}, stride={self.stride}, '
                f'padding={self.padding}, bias={self.bias is not None}')


class GroupedConv2d(nn.Module):
    """nn.Conv2d(..., groups=G > 1)'s parameters and keys: ``weight`` (Cout,
    Cin / G, k, k).  FORWARD ONLY -- it exists for the frozen ResNeXt teacher of
    BASELINE config 5 (resnext.py:49-61: the 3x3 conv2 of every Bottleneck,
    groups = 32); runs ld_gconv_forward with the following eval-mode BN and ReLU
    folded into its epilogue (``forward3_fused``, the hook resnet._conv_bn
    takes for forward-only convs)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias=False):
        super().__init__()
        k, s, p, d = (_pair(kernel_size), _pair(stride), _pair(padding),
                      _pair(dilation))
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1]:
            raise NotImplementedError('only square kernels/strides/pads')
        if d != (1, 1) or bias:
            raise NotImplementedError('grouped conv: no dilation, no bias '
                                      '(resnext.py:49-61)')
        if in_channels % groups or out_channels % groups or \
                out_channels // groups not in (4, 8, 16, 32) or \
                k[0] not in (1, 3):
            raise NotImplementedError(
                f'grouped conv {in_channels}->{out_channels} / {groups} k{k[0]}: '
                'built for 4 / 8 / 16 / 32 output channels per group, k 1 or 3')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = k, s, p
        self.groups = groups
        self.weight = nn.Parameter(
            torch.empty(out_channels, in_channels // groups, k[0], k[1]))
        self.bias = None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward3_fused(self, x3, levels, scale=None, shift=None, residual=None,
                       relu=False):
        if torch.is_grad_enabled() and (x3.requires_grad or
                                        self.weight.requires_grad):
            raise NotImplementedError(
                'grouped convs are forward-only here (the frozen ResNeXt '
                'teacher of config 5): call them under torch.no_grad()')
        if residual is not None:
            raise NotImplementedError('grouped conv with a fused residual')
        return Y.gconv_forward(x3, self.weight, self.groups, self.stride[0],
                               self.padding[0], levels, scale, shift, relu)

    def forward3(self, x3, levels):
        return self.forward3_fused(x3, levels)

    def forward(self, x):
        n, c, h, w = x.shape
        y3, lv = self.forward3(x.reshape(n, c, h * w), ((h, w), ))
        return y3.view(n, self.out_channels, lv[0][0], lv[0][1])

    def extra_repr(self):
        return (f'{self.in_channels}, {self.out_channels}, '
                f'kernel_size={self.kernel_size}, stride={self.stride}, '
                f'padding={self.padding}, groups={self.groups}, bias=False')


class DeformConv2dPack(nn.Module):
    """mmcv.ops.DeformConv2dPack (conv type 'DCN'): deformable convolution v1
    whose offsets come from its own ``conv_offset`` 3x3 conv (zero-initialised,
    with bias).  Parameter names / shapes as mmcv's: ``weight`` (Cout, Cin, k,
    k), no bias, ``conv_offset.weight`` (2*k*k*deform_groups, Cin, k, k),
    ``conv_offset.bias``.  FORWARD ONLY: it exists for the frozen R101-DCN
    teacher of config 4 (resnet.py:171-194); deform_groups = groups = 1."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, deform_groups=1, bias=False,
                 **kwargs):
        super().__init__()
        k, s, p, d = (_pair(kernel_size), _pair(stride), _pair(padding),
                      _pair(dilation))
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or d[0] != d[1]:
            raise NotImplementedError('only square kernels/strides/pads')
        if deform_groups != 1 or bias:
            raise NotImplementedError(
                'DCN: deform_groups = 1 and no bias (the configs/gfl/*dconv* '
                'and configs/imv2/gflv2_x101* settings) are built')
        if groups != 1 and (in_channels % groups or out_channels % groups or
                            out_channels // groups not in (4, 8, 16, 32)):
            raise NotImplementedError(
                f'grouped DCN {in_channels}->{out_channels} / {groups}')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = k, s, p, d
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(
            torch.empty(out_channels, in_channels // groups, k[0], k[1]))
        self.bias = None
        self.conv_offset = Conv2d(in_channels, deform_groups * 2 * k[0] * k[1],
                                  k[0], stride=s[0], padding=p[0], bias=True)
        if d != (1, 1):
            raise NotImplementedError('dilated DCN')
        self.reset_parameters()
        self._w2d = None

    def reset_parameters(self):
        # mmcv: uniform(-stdv, stdv), stdv = 1/sqrt(Cin*k*k); offsets start at 0
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        nn.init.uniform_(self.weight, -stdv, stdv)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def _weight_2d(self):
        """(Cout, Cin*k*k, 1, 1) view of the weight, kept alive so the GEMM
        weight image cached on it survives between calls."""
        w, v = self.weight, self._w2d
        if v is None or v.data_ptr() != w.data_ptr() or v.device != w.device \
                or v._version != w._version:
            v = w.detach().view(self.out_channels, -1, 1, 1)
            v._ld_static = getattr(w, '_ld_static', False) or \
                not w.requires_grad
            self._w2d = v
        return v

    def forward3_fused(self, x3, levels, scale=None, shift=None, residual=None,
                       relu=False):
        """offset conv -> deformable im2col -> 1x1 GEMM with the fused epilogue
        (BN affine / residual / ReLU)."""
        if torch.is_grad_enabled() and (x3.requires_grad or
                                        self.weight.requires_grad):
            raise NotImplementedError(
                'DeformConv2dPack is forward-only here (the frozen teacher of '
                'config 4): call it under torch.no_grad()')
        if len(levels) != 1:
            raise NotImplementedError('DCN on level-concatenated tensors')
        (h, w), = levels
        N, cin, P = x3.shape
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        off3, out_levels = Y.conv_forward_raw(
            x3, self.conv_offset.weight, s, p, levels,
            bias=self.conv_offset.bias)
        (ho, wo), = out_levels
        col = Y.deform_im2col(x3, off3, h, w, k, s, p, 1)
        if self.groups != 1:
            # grouped DCN (ResNeXt-DCN, resnext.py:62-74): rows [g * cg * k * k,
            # (g + 1) * cg * k * k) of the column tensor feed group g -- a grouped
            # 1x1 conv over Cin * k * k channels
            if residual is not None:
                raise NotImplementedError('grouped DCN with a fused residual')
            y3, _ = Y.gconv_forward(col, self._weight_2d(), self.groups, 1, 0,
                                    ((ho, wo), ), scale, shift, relu)
            return y3, out_levels
        y3, _ = Y.conv_forward_raw(col, self._weight_2d(), 1, 0,
                                   ((ho, wo), ), scale=scale, shift=shift,
                                   residual=residual, relu=relu)
        return y3, out_levels

    def forward3(self, x3, levels):
        return self.forward3_fused(x3, levels)

    def forward(self, x):
        n, c, h, w = x.shape
        y3, lv = self.forward3(x.reshape(n, c, h * w), ((h, w), ))
        return y3.view(n, self.out_channels, lv[0][0], lv[0][1])


class BatchNorm2d(nn.Module):
    """BatchNorm2d evaluated with its running statistics (the LD configs run
    every BN with norm_eval=True, resnet.py:639-648); affine stays trainable.
    Training-mode batch statistics are not on this path and raise."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked',
                             torch.tensor(0, dtype=torch.long))

    def forward3(self, x3, residual=None, relu=False):
        if self.training:
            raise NotImplementedError(
                'BatchNorm2d in training mode (batch statistics) is outside '
                'the LD hot path: use norm_eval=True')
        return Y.bn_act(x3, self.weight, self.bias, self.running_mean,
                        self.running_var, self.eps, residual, relu)

    def forward(self, x):
        n, c, h, w = x.shape
        return self.forward3(x.reshape(n, c, h * w)).view(n, c, h, w)


class GroupNorm(nn.Module):

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = (num_groups,
                                                        num_channels, eps)
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward3(self, x3, levels, relu=False):
        return Y.gn_act(x3, self.weight, self.bias, self.num_groups, self.eps,
                        levels, relu)

    def forward(self, x):
        n, c, h, w = x.shape
        return self.forward3(x.reshape(n, c, h * w), ((h, w), )).view(
            n, c, h, w)


class ReLU(nn.Module):
    """Marker module (mmcv ConvModule.activate); the ReLU itself is fused
    into the preceding norm / conv epilogue kernel."""

    def __init__(self, inplace=True):
        super().__init__()
        self.inplace = inplace


def build_conv_layer(cfg, *args, **kwargs):
    if cfg is None:
        cfg = dict(type='Conv2d')
    layer_type = cfg['type'] if isinstance(cfg, dict) else cfg
    if layer_type in ('Conv2d', 'Conv'):
        if kwargs.get('groups', 1) != 1:
            return GroupedConv2d(*args, **kwargs)
        return Conv2d(*args, **kwargs)
    if layer_type == 'DCN':
        kw = {k: v for k, v in cfg.items() if k != 'type'}
        kw.update(kwargs)
        return DeformConv2dPack(*args, **kw)
    raise NotImplementedError(
        f'conv layer type {layer_type} is not implemented on the MI355X path '
        '(DCNv2 / grouped convs are outside SURVEY.md section 8)')


def build_norm_layer(cfg, num_features, postfix=''):
    """-> (name, layer); mmcv: 'BN' -> bn{postfix}, 'GN' -> gn{postfix}."""
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type == 'BN':
        layer, abbr = BatchNorm2d(num_features, **cfg_), 'bn'
    elif layer_type == 'GN':
        assert 'num_groups' in cfg_
        layer, abbr = GroupNorm(num_channels=num_features, **cfg_), 'gn'
    else:
        raise NotImplementedError(f'norm type {layer_type}')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        n, c, h, w = x.shape
        return Y.scale_levels(x.reshape(n, c, h * w), self.scale.reshape(1),
                              ((h, w), )).view(n, c, h, w)


class ConvModule(nn.Module):
    """conv -> norm -> activation, mmcv semantics: bias='auto' means bias iff
    there is no norm; the norm layer is registered as ``gn`` / ``bn``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias='auto', conv_cfg=None,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 with_spectral_norm=False, padding_mode='zeros',
                 order=('conv', 'norm', 'act')):
        super().__init__()
        assert order == ('conv', 'norm', 'act') and padding_mode == 'zeros'
        assert not with_spectral_norm
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels,
                                     kernel_size, stride=stride,
                                     padding=padding, dilation=dilation,
                                     groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            if act_cfg['type'] != 'ReLU':
                raise NotImplementedError(act_cfg['type'])
            self.activate = ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def forward3(self, x3, levels, activate=True, norm=True, c8_out=False):
        """``c8_out``: the caller's next layer is another conv of this kind -- a
        frozen conv + GN layer in bf16 mode may then return its output as a
        layers.C8Act (no fp32 copy)."""
        act = activate and self.with_activation
        if act and not (norm and self.with_norm):
            # conv (+bias) -> ReLU (RetinaGFLHead's towers, norm_cfg=None)
            c = self.conv
            if type(c) is Conv2d and not (torch.is_grad_enabled() and (
                    x3.requires_grad or c.weight.requires_grad or
                    (c.bias is not None and c.bias.requires_grad))):
                return Y.conv_forward_raw(x3, c.weight, c.stride[0],
                                          c.padding[0], levels, bias=c.bias,
                                          relu=True, emit_c8=True)
            y3, lv = c.forward3(x3, levels)
            return Y.relu(y3), lv
        if norm and self.with_norm and type(self.conv) is Conv2d and \
                self.conv.bias is None and type(self.norm) is GroupNorm:
            # trainable tower layer: one autograd node in bf16 mode (the pair
            # of launches otherwise) -- layers.conv_gn_act
            c, n = self.conv, self.norm
            return Y.conv_gn_act(x3, c.weight, n.weight, n.bias, n.num_groups,
                                 n.eps, c.stride[0], c.padding[0], levels,
                                 relu=act, c8_only=c8_out)
        y3, lv = self.conv.forward3(x3, levels)
        if norm and self.with_norm:
            n = self.norm
            if isinstance(n, GroupNorm):
                y3 = n.forward3(y3, lv, relu=act)
            else:
                y3 = n.forward3(y3, None, relu=act)
        return y3, lv

    def forward(self, x, activate=True, norm=True):
        n, c, h, w = x.shape
        y3, lv = self.forward3(x.reshape(n, c, h * w), ((h, w), ), activate,
                               norm)
        return y3.view(n, self.out_channels, lv[0][0], lv[0][1])
