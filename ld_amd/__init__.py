"""ld_amd -- MI355X-native implementation of the HikariTJU/LD training hot
path (student+teacher GFocal dual forward, ATSS/VLR/IM targets, the fused LD
loss block, backward, gradient all-reduce, SGD) behind mmdet's registry API.

Importing the package registers every component under its mmdet ``type``
name; nothing here runs on the CPU -- ops raise ``LdError`` when handed a
non-HIP tensor or when libldhip.so has not been built.
"""
from . import registry  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .lib import LdError  # noqa: F401
from . import core, losses, resnet, fpn, heads, detectors  # noqa: F401,E402
from .registry import (build_backbone, build_detector, build_head,  # noqa
                       build_loss, build_neck)

__version__ = '0.1.0'
