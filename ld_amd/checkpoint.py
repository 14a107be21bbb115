"""Checkpoint plumbing (mmcv.runner.load_checkpoint subset): state_dict files
with mmdet key names.  URL schemes (torchvision://, https://) need network
access, which this environment does not have -> they raise."""
import os

import torch


def load_checkpoint(model, filename, map_location='cpu', strict=False,
                    logger=None):
    if not isinstance(filename, str):
        raise TypeError('filename must be a str')
    if '://' in filename or not os.path.isfile(filename):
        raise FileNotFoundError(
            f'checkpoint {filename!r} is not a local file (no network access: '
            'torchvision:// and https:// checkpoints cannot be fetched)')
    ckpt = torch.load(filename, map_location=map_location)
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {k[7:] if k.startswith('module.') else k: v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    return dict(missing=missing, unexpected=unexpected, meta=ckpt.get(
        'meta', {}) if isinstance(ckpt, dict) else {})
