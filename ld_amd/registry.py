"""Registry / build_from_cfg: the drop-in boundary of the reference
(mmdet/models/builder.py:6-77, mmdet/core/bbox/builder.py:3-20,
mmdet/core/anchor/builder.py:3-7; mmcv.utils.Registry semantics).

Components are looked up by their mmdet ``type`` string, so the reference's
``configs/ld/*.py`` (and the ``configs/gfl/*`` teacher configs they point to)
resolve against this package unchanged.
"""
import inspect

__all__ = [
    'Registry', 'build_from_cfg', 'BACKBONES', 'NECKS', 'HEADS', 'LOSSES',
    'DETECTORS', 'BBOX_ASSIGNERS', 'BBOX_SAMPLERS', 'BBOX_CODERS',
    'ANCHOR_GENERATORS', 'IOU_CALCULATORS', 'build_backbone', 'build_neck',
    'build_head', 'build_loss', 'build_detector', 'build_assigner',
    'build_sampler', 'build_bbox_coder', 'build_anchor_generator',
    'build_iou_calculator'
]


class Registry:

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, got {type(module_class)}')
        names = module_name or module_class.__name__
        if isinstance(names, str):
            names = [names]
        for name in names:
            if not force and name in self._module_dict:
                raise KeyError(f'{name} is already registered in {self.name}')
            self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, got {type(force)}')
        if module is not None:
            self._register_module(module, name, force)
            return module

        def _register(cls):
            self._register_module(cls, name, force)
            return cls

        return _register


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(
                f'`cfg` or `default_args` must contain the key "type", got '
                f'{cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry, got {type(registry)}')
    args = dict(cfg)
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(
                f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or class, got {type(obj_type)}')
    return obj_cls(**args)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
ANCHOR_GENERATORS = Registry('Anchor generator')
IOU_CALCULATORS = Registry('IoU calculator')


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        import torch.nn as nn
        return nn.Sequential(
            *[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:67-77."""
    assert cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field'
    assert cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field'
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)
