"""Config.fromfile for mmdet-style Python configs: ``_base_`` inheritance
(relative paths, lists), recursive dict merge, ``_delete_``, attribute access.
Enough of mmcv.Config for the reference's configs/ld, configs/ldv2,
configs/gfl and configs/_base_ files (tools/train.py:89-95).
"""
import os
import types

__all__ = ['Config', 'ConfigDict']


class ConfigDict(dict):

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(
                f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    @classmethod
    def wrap(cls, obj):
        if isinstance(obj, dict):
            return cls({k: cls.wrap(v) for k, v in obj.items()})
        if isinstance(obj, list):
            return [cls.wrap(v) for v in obj]
        if isinstance(obj, tuple):
            return tuple(cls.wrap(v) for v in obj)
        return obj


def _merge_a_into_b(a, b):
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get('_delete_', False):
            if not isinstance(b[k], dict):
                raise TypeError(
                    f'{k}={v} in child config cannot inherit from base '
                    f'because {k} is a dict in the child config but is of '
                    f'type {type(b[k])} in base config. You may set '
                    '`_delete_=True` to ignore the base config')
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            b[k] = v
    return b


class Config:

    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is None:
            cfg_dict = {}
        elif not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, got {type(cfg_dict)}')
        object.__setattr__(self, '_cfg_dict', ConfigDict.wrap(cfg_dict))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(f'config file {filename} does not exist')
        if not filename.endswith('.py'):
            raise IOError('Only py type configs are supported')
        scope = {'__file__': filename}
        with open(filename) as f:
            exec(compile(f.read(), filename, 'exec'), scope)
        cfg = {
            k: v
            for k, v in scope.items()
            if not k.startswith('__') and not isinstance(
                v, (types.ModuleType, types.FunctionType))
        }
        if '_base_' in cfg:
            base = cfg.pop('_base_')
            base = base if isinstance(base, list) else [base]
            merged = {}
            for b in base:
                sub = Config._file2dict(
                    os.path.join(os.path.dirname(filename), b))
                dup = set(merged) & set(sub)
                if dup:
                    raise KeyError(f'Duplicate key is not allowed among bases: '
                                   f'{sorted(dup)}')
                merged.update(sub)
            cfg = _merge_a_into_b(cfg, merged)
        return cfg

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def merge_from_dict(self, options):
        """``--cfg-options a.b=1`` style overrides (tools/train.py:90-91)."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split('.')
            for sub in keys[:-1]:
                d = d.setdefault(sub, {})
            d[keys[-1]] = v
        object.__setattr__(
            self, '_cfg_dict',
            ConfigDict.wrap(_merge_a_into_b(nested, self._cfg_dict)))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __repr__(self):
        return f'Config (path: {self._filename}): {dict(self._cfg_dict)!r}'
