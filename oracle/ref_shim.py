"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ld_amd`.

Import shim that lets the *unmodified* reference package at /root/reference
(`mmdet`, an MMDetection 2.10 fork) import and run on CPU inside this container,
where its third-party dependencies (mmcv-full 1.2.x, pycocotools, cv2,
torchvision, terminaltables) are absent.

It is used in exactly one place: `oracle/gen_golden.py`, which executes the
reference code to produce the golden vectors committed under `tests/golden/`.
It cannot travel to the GPU box (/root/reference does not exist there).

What is REAL here (restated from mmcv 1.2.x's public behaviour, the arithmetic
the reference relies on -- SURVEY.md section 8c):
  Registry / build_from_cfg / ConfigDict / Config.fromfile (with _base_),
  ConvModule (conv -> norm -> act, bias='auto'), build_conv_layer,
  build_norm_layer (BN / GN), Scale, the *_init helpers, force_fp32/auto_fp16
  (identity), mmcv.jit (identity), mmcv.ops.nms / batched_nms (greedy NMS).
Everything else reachable under the fabricated roots resolves to permissive
dummies so `import mmdet` succeeds.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get('LD_REFERENCE_ROOT', '/root/reference')
_FAKE_CANDIDATES = ('mmcv', 'pycocotools', 'terminaltables', 'cv2',
                    'torchvision', 'xdoctest', 'matplotlib', 'seaborn')
# only fabricate what is genuinely absent from this interpreter
_FAKE_ROOTS = tuple(
    r for r in _FAKE_CANDIDATES if importlib.util.find_spec(r) is None)


# --------------------------------------------------------------------------
# real pieces
# --------------------------------------------------------------------------
class ConfigDict(dict):
    """Attribute-style dict (mmcv.utils.ConfigDict behaviour we need)."""

    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError:
            raise AttributeError(name)
        return v

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return ConfigDict({k: ConfigDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(ConfigDict.wrap(v) for v in obj)
        return obj


class Registry:

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register(self, cls, name=None, force=False):
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError(f'{name} already registered in {self._name}')
        self._module_dict[name] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _dec(cls):
            self._register(cls, name, force)
            return cls

        return _dec


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    else:
        cls = obj_type
    return cls(**args)


def _merge(base, over):
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get(
                '_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = v
    return out


class Config:

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg', ConfigDict.wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _load(path):
        path = os.path.abspath(path)
        scope = {}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), scope)
        cfg = {
            k: v
            for k, v in scope.items()
            if not k.startswith('__') and not isinstance(v, types.ModuleType)
            and not callable(v)
        }
        bases = cfg.pop('_base_', None)
        if bases is not None:
            if isinstance(bases, str):
                bases = [bases]
            merged = {}
            for b in bases:
                merged = _merge(
                    merged, Config._load(os.path.join(os.path.dirname(path), b)))
            cfg = _merge(merged, cfg)
        return cfg

    @staticmethod
    def fromfile(path):
        return Config(Config._load(path), filename=path)

    def __getattr__(self, name):
        return getattr(self._cfg, name)

    def __getitem__(self, name):
        return self._cfg[name]

    def get(self, k, d=None):
        return self._cfg.get(k, d)


def _identity_decorator_factory(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def _dec(fn):
        return fn

    return _dec


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module,
                 a=0,
                 mode='fan_out',
                 nonlinearity='relu',
                 bias=0,
                 distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(
                module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(
                module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


def build_conv_layer(cfg, *args, **kwargs):
    if cfg is None:
        cfg = dict(type='Conv2d')
    t = cfg['type'] if isinstance(cfg, dict) else cfg
    if t not in ('Conv2d', 'Conv'):
        raise NotImplementedError(f'oracle shim: conv type {t} not available')
    return nn.Conv2d(*args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if t == 'BN':
        layer, abbr = nn.BatchNorm2d(num_features, **cfg), 'bn'
    elif t == 'GN':
        layer, abbr = nn.GroupNorm(
            num_channels=num_features, **cfg), 'gn'
    else:
        raise NotImplementedError(f'oracle shim: norm type {t}')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


class Scale(nn.Module):

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


class ConvModule(nn.Module):
    """conv -> norm -> act; bias='auto' means bias iff there is no norm."""

    def __init__(self,
                 in_channels,
                 out_channels,
                 kernel_size,
                 stride=1,
                 padding=0,
                 dilation=1,
                 groups=1,
                 bias='auto',
                 conv_cfg=None,
                 norm_cfg=None,
                 act_cfg=dict(type='ReLU'),
                 inplace=True,
                 **kwargs):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(
            conv_cfg,
            in_channels,
            out_channels,
            kernel_size,
            stride=stride,
            padding=padding,
            dilation=dilation,
            groups=groups,
            bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(getattr(self, self.norm_name), 1, bias=0)

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def forward(self, x, activate=True, norm=True):
        x = self.conv(x)
        if norm and self.with_norm:
            x = self.norm(x)
        if activate and self.with_activation:
            x = self.activate(x)
        return x


def is_tuple_of(seq, expected_type):
    return isinstance(seq, tuple) and all(
        isinstance(s, expected_type) for s in seq)


def _load_checkpoint(*a, **k):
    raise RuntimeError('oracle shim: checkpoints are not available offline')


# --------------------------------------------------------------------------
# mmcv.ops.nms (mmcv-full 1.2.x; a compiled op there) -- restated.
# nms: boxes sorted by score (descending), greedy suppression of every later box
# with IoU > iou_threshold, IoU = inter / (area_i + area_j - inter) with
# `offset` added to widths/heights (0 by default).  The compiled op sorts with
# an unstable sort; equal scores are ordered lower-index-first here.
# batched_nms: per-class NMS by shifting each class's boxes by
# label * (max_coordinate + 1); with >= split_thr boxes the classes are run one
# by one and the survivors re-sorted by score.
# --------------------------------------------------------------------------
def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0)
    if boxes.numel() == 0:
        keep = boxes.new_zeros((0, ), dtype=torch.long)
        return torch.cat([boxes, scores[:, None]], -1), keep
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1 + offset) * (y2 - y1 + offset)
    n = b.size(0)
    removed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        xx1 = torch.maximum(x1[i], x1[i + 1:])
        yy1 = torch.maximum(y1[i], y1[i + 1:])
        xx2 = torch.minimum(x2[i], x2[i + 1:])
        yy2 = torch.minimum(y2[i], y2[i + 1:])
        w = (xx2 - xx1 + offset).clamp(min=0)
        h = (yy2 - yy1 + offset).clamp(min=0)
        inter = w * h
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        removed[i + 1:] |= ovr > iou_threshold
    keep = order[torch.tensor(keep, dtype=torch.long)]
    if max_num > 0:
        keep = keep[:max_num]
    return torch.cat([boxes[keep], scores[keep][:, None]], -1), keep


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_type = nms_cfg_.pop('type', 'nms')
    assert nms_type == 'nms', nms_type
    split_thr = nms_cfg_.pop('split_thr', 10000)
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms(boxes_for_nms, scores, **nms_cfg_)
        boxes = boxes[keep]
        scores = dets[:, -1]
    else:
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        for id in torch.unique(idxs):
            mask = (idxs == id).nonzero(as_tuple=False).view(-1)
            dets, keep = nms(boxes_for_nms[mask], scores[mask], **nms_cfg_)
            total_mask[mask[keep]] = True
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        keep = keep[scores[keep].argsort(descending=True, stable=True)]
        boxes = boxes[keep]
        scores = scores[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


_REAL = {
    'mmcv.ops': dict(nms=nms, batched_nms=batched_nms),
    'mmcv.ops.nms': dict(nms=nms, batched_nms=batched_nms),
    'mmcv': dict(
        __version__='1.2.7', Config=Config, ConfigDict=ConfigDict,
        jit=_identity_decorator_factory, is_tuple_of=is_tuple_of),
    'mmcv.utils': dict(
        Registry=Registry, build_from_cfg=build_from_cfg, Config=Config,
        ConfigDict=ConfigDict, is_tuple_of=is_tuple_of,
        print_log=lambda *a, **k: None),
    'mmcv.runner': dict(
        force_fp32=_identity_decorator_factory,
        auto_fp16=_identity_decorator_factory,
        load_checkpoint=_load_checkpoint,
        HOOKS=Registry('hook'), Hook=type('Hook', (), {}),
        get_dist_info=lambda: (0, 1)),
    'mmcv.runner.hooks': dict(HOOKS=Registry('hook'),
                              Hook=type('Hook', (), {})),
    'mmcv.cnn': dict(
        ConvModule=ConvModule, Scale=Scale, build_conv_layer=build_conv_layer,
        build_norm_layer=build_norm_layer, normal_init=normal_init,
        constant_init=constant_init, xavier_init=xavier_init,
        kaiming_init=kaiming_init, bias_init_with_prob=bias_init_with_prob),
}


# --------------------------------------------------------------------------
# permissive fabricated modules
# --------------------------------------------------------------------------
def _dummy_callable(*args, **kwargs):
    # unknown decorator used bare -> identity; used with args -> itself
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return _dummy_callable


class _FakeModule(types.ModuleType):

    def __getattr__(self, name):
        if name == '__version__':  # mmdet/datasets/coco.py:21 compares it
            return '99.0.0'
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        if name[:1].isupper():
            cls = type(name, (nn.Module, ), {
                '__init__': lambda self, *a, **k: nn.Module.__init__(self),
                '__module__': self.__name__
            })
            setattr(self, name, cls)
            return cls
        return _dummy_callable


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in _FAKE_ROOTS or \
                fullname == 'numpy.lib.twodim_base':
            return importlib.machinery.ModuleSpec(fullname, self,
                                                  is_package=True)
        return None

    def create_module(self, spec):
        m = _FakeModule(spec.name)
        m.__path__ = []
        for k, v in _REAL.get(spec.name, {}).items():
            setattr(m, k, v)
        if spec.name == 'numpy.lib.twodim_base':
            m.tri = np.tri
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Make `import mmdet` resolve to the reference under the shim."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, 'mmdet')):
        raise RuntimeError(
            f'reference not present at {REFERENCE_ROOT}: the oracle shim only '
            'works in the build container')
    # numpy>=2 has no numpy.lib.twodim_base (kd_one_stage.py:2 imports it)
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
