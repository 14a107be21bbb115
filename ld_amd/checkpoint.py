"""Checkpoint wire format.

The reference reads and writes checkpoints through ``mmcv.runner`` (a
dependency that is not vendored in the reference tree: mmcv-full, pinned by
``mmdet/__init__.py:18-19`` to 1.2.4 <= v <= 1.3).  Its published layout,
restated here, is one ``torch.save``d dict

    {'meta':       {'mmcv_version', 'time', + caller meta: tools/train.py:168-173
                    adds 'mmdet_version', 'CLASSES'; EpochBasedRunner adds
                    'epoch', 'iter'},
     'state_dict': OrderedDict[str, CPU tensor]   (DDP's 'module.' stripped),
     'optimizer':  torch.optim.Optimizer.state_dict()}

and these call sites are the contract on the LD path:
  * ``load_checkpoint(self.teacher_model, teacher_ckpt, map_location='cpu')``
    (detectors/kd_one_stage.py:42-44): a released GFL teacher ``.pth``;
  * ``load_checkpoint(self, pretrained, strict=False)`` in
    ``ResNet.init_weights`` (backbones/resnet.py:597-599) with
    ``pretrained='torchvision://resnet50'`` (configs/ld/*.py): torchvision's
    key names == mmdet's ResNet key names, ``fc.*`` is reported unexpected;
  * ``save_checkpoint(model, filename, optimizer=..., meta=...)`` from the
    runner's CheckpointHook: the teacher is NOT in the student's state_dict
    (kd_one_stage.py:97-108 hides it from ``nn.Module`` registration).

There is no network here: URL schemes are resolved against local directories
(``LD_CHECKPOINT_DIR``, ``$TORCH_HOME/hub/checkpoints``,
``~/.cache/torch/hub/checkpoints``) by the file name the scheme would download
to, and raise ``FileNotFoundError`` naming that file otherwise.

The optimizer entry is interchangeable with ``torch.optim.SGD`` although the
trainer keeps momentum in one flat arena: ``SGDTrainer.state_dict()`` emits the
per-parameter ``momentum_buffer`` layout torch would (parameter index = position
in ``model.parameters()``, frozen ones included, as mmcv's
DefaultOptimizerConstructor passes all of them in one group).
"""
import os
import time
import warnings
from collections import OrderedDict

import torch

MMCV_VERSION = '1.2.7'  # inside the window the reference pins (1.2.4 .. 1.3)

# file names torchvision.models.<arch>.model_urls download to (the targets of
# mmcv's ``torchvision://`` scheme)
_TORCHVISION_FILES = {
    'resnet18': 'resnet18-5c106cde.pth',
    'resnet34': 'resnet34-333f7ec4.pth',
    'resnet50': 'resnet50-19c8e357.pth',
    'resnet101': 'resnet101-5d3b4d8f.pth',
    'resnet152': 'resnet152-b121ed2d.pth',
    'resnext50_32x4d': 'resnext50_32x4d-7cdf4587.pth',
    'resnext101_32x8d': 'resnext101_32x8d-8ba56ff5.pth',
}


def _search_dirs():
    dirs = []
    if os.environ.get('LD_CHECKPOINT_DIR'):
        dirs.append(os.environ['LD_CHECKPOINT_DIR'])
    home = os.environ.get('TORCH_HOME',
                          os.path.join(os.path.expanduser('~'), '.cache',
                                       'torch'))
    dirs.append(os.path.join(home, 'hub', 'checkpoints'))
    dirs.append(os.path.join(home, 'checkpoints'))
    return dirs


def resolve_checkpoint_path(filename):
    """Local path of a checkpoint reference (plain path, ``torchvision://arch``,
    ``open-mmlab://...``, ``http(s)://...``)."""
    if not isinstance(filename, str):
        raise TypeError('filename must be a str')
    if '://' not in filename:
        if not os.path.isfile(filename):
            raise FileNotFoundError(f'{filename} is not a checkpoint file')
        return filename
    scheme, rest = filename.split('://', 1)
    if scheme == 'torchvision':
        if rest not in _TORCHVISION_FILES:
            raise KeyError(f'torchvision://{rest}: unknown architecture '
                           f'(known: {sorted(_TORCHVISION_FILES)})')
        base = _TORCHVISION_FILES[rest]
    else:
        base = os.path.basename(rest)
    for d in _search_dirs():
        cand = os.path.join(d, base)
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError(
        f'{filename}: no network access here; place {base!r} in one of '
        f'{_search_dirs()} (LD_CHECKPOINT_DIR overrides)')


def _strip_prefix(state_dict, prefix='module.'):
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k[len(prefix):] if k.startswith(prefix) else k] = v
    return out


def load_state_dict(module, state_dict, strict=False, logger=None):
    """mmcv.runner.load_state_dict: copy matching entries, REPORT (not raise,
    unless ``strict``) missing / unexpected / shape-mismatched keys.
    ``num_batches_tracked`` buffers are not counted as missing."""
    own = module.state_dict()
    unexpected = [k for k in state_dict if k not in own]
    missing = [k for k in own
               if k not in state_dict and 'num_batches_tracked' not in k]
    mismatched = []
    with torch.no_grad():
        for k, v in state_dict.items():
            if k not in own:
                continue
            if tuple(own[k].shape) != tuple(v.shape):
                mismatched.append((k, tuple(v.shape), tuple(own[k].shape)))
                continue
            own[k].copy_(v)
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' +
                    ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    for k, a, b in mismatched:
        msgs.append(f'size mismatch for {k}: checkpoint {a} vs model {b}')
    if msgs:
        text = 'The model and loaded state dict do not match exactly\n' + \
            '\n'.join(msgs)
        if strict:
            raise RuntimeError(text)
        if logger is not None:
            logger.warning(text)
        else:
            warnings.warn(text)
    return dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def load_checkpoint(model, filename, map_location='cpu', strict=False,
                    logger=None):
    """mmcv.runner.load_checkpoint: returns the checkpoint dict."""
    path = resolve_checkpoint_path(filename)
    ckpt = torch.load(path, map_location=map_location)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    report = load_state_dict(model, _strip_prefix(sd), strict, logger)
    if 'state_dict' not in ckpt:  # bare state_dict file (torchvision zoo)
        ckpt = dict(state_dict=sd, meta={})
    ckpt['_load_report'] = report
    return ckpt


def get_state_dict(model):
    """Student state_dict with mmdet key names on the CPU.  DDP-style wrappers
    (``.module``) are unwrapped; the teacher is not a registered sub-module and
    so never appears."""
    inner = getattr(model, 'module', model)
    return OrderedDict((k, v.detach().cpu())
                       for k, v in inner.state_dict().items())


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """mmcv.runner.save_checkpoint layout.  ``optimizer`` is anything with a
    ``state_dict()`` (``SGDTrainer`` emits torch.optim.SGD's format)."""
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError(f'meta must be a dict or None, but got {type(meta)}')
    meta = dict(meta)
    meta.update(mmcv_version=MMCV_VERSION, time=time.asctime())
    inner = getattr(model, 'module', model)
    if getattr(inner, 'CLASSES', None) is not None:
        meta.update(CLASSES=inner.CLASSES)
    ckpt = dict(meta=meta, state_dict=get_state_dict(model))
    if optimizer is not None:
        sd = optimizer.state_dict()
        ckpt['optimizer'] = sd
    d = os.path.dirname(os.path.abspath(filename))
    os.makedirs(d, exist_ok=True)
    tmp = filename + '.tmp'
    with open(tmp, 'wb') as f:
        torch.save(ckpt, f)
        f.flush()
    os.replace(tmp, filename)
    return ckpt


def resume(trainer, filename, map_location='cpu', resume_optimizer=True):
    """BaseRunner.resume: model + optimizer + (epoch, iter) from ``meta``."""
    ckpt = load_checkpoint(trainer.model, filename, map_location=map_location)
    if resume_optimizer and 'optimizer' in ckpt:
        trainer.load_state_dict(ckpt['optimizer'])
    meta = ckpt.get('meta', {})
    trainer.iter = int(meta.get('iter', trainer.iter))
    trainer.epoch = int(meta.get('epoch', getattr(trainer, 'epoch', 0)))
    return ckpt
