"""Reference-side binding: import this module from an mmdet config
(``custom_imports = dict(imports=['ld_amd.mmdet_plugin'])``, reference
tools/train.py:93-95) to make every ``type=`` name of configs/ld/* resolve to
the MI355X implementation inside an existing mmdet 2.10 installation.

Needs mmdet + mmcv importable (they are not in this repository's image; the
stand-alone mode -- ld_amd.Config / ld_amd.build_detector -- needs neither).
"""
import ld_amd
from ld_amd import registry as R

try:
    from mmdet.core.anchor.builder import ANCHOR_GENERATORS
    from mmdet.core.bbox.builder import (BBOX_ASSIGNERS, BBOX_CODERS,
                                         BBOX_SAMPLERS)
    from mmdet.core.bbox.iou_calculators.builder import IOU_CALCULATORS
    from mmdet.models.builder import (BACKBONES, DETECTORS, HEADS, LOSSES,
                                      NECKS)
except ImportError as e:  # pragma: no cover - depends on the host install
    raise ImportError(
        'ld_amd.mmdet_plugin re-registers the HIP components over mmdet\'s '
        'registries and therefore needs mmdet/mmcv; use ld_amd.build_detector '
        'for the stand-alone mode') from e

_PAIRS = [(BACKBONES, R.BACKBONES), (NECKS, R.NECKS), (HEADS, R.HEADS),
          (LOSSES, R.LOSSES), (DETECTORS, R.DETECTORS),
          (BBOX_ASSIGNERS, R.BBOX_ASSIGNERS), (BBOX_SAMPLERS, R.BBOX_SAMPLERS),
          (BBOX_CODERS, R.BBOX_CODERS),
          (ANCHOR_GENERATORS, R.ANCHOR_GENERATORS),
          (IOU_CALCULATORS, R.IOU_CALCULATORS)]

for theirs, ours in _PAIRS:
    for name, cls in ours.module_dict.items():
        theirs.register_module(name=name, force=True, module=cls)

__all__ = ['ld_amd']
