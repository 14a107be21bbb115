"""ld_amd -- MI355X-native implementation of the HikariTJU/LD training hot
path (student+teacher GFocal dual forward, ATSS/VLR/IM targets, the fused LD
loss block, backward, gradient all-reduce, SGD) behind mmdet's registry API.

Importing the package registers every component under its mmdet ``type``
name; nothing here runs on the CPU -- ops raise ``LdError`` when handed a
non-HIP tensor or when libldhip.so has not been built.
"""
import os as _os

# Three HIP streams carry the train step (student, teacher one step ahead,
# weight gradients) and RCCL adds its own; the ROCm runtime multiplexes streams
# onto 4 hardware queues by default, and with a process group present two of the
# step's streams share one queue (measured: 36.8 vs 35.1 ms per step).  More
# queues fix that -- but make hipGraph replays much slower (bf16 15.2 -> 26 ms),
# so the value is raised only in a multi-process job (WORLD_SIZE > 1), whose
# steps AutoStepper enqueues eagerly.  The runtime reads it
# when it initialises (the first HIP call): set at import, unless the
# application chose a value itself; ld_amd.train warns when it came too late.
if int(_os.environ.get('WORLD_SIZE', '1') or 1) > 1 or \
        _os.environ.get('LD_FORCE_COLLECTIVES') == '1':
    _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from . import registry  # noqa: F401,E402
from .config import Config, ConfigDict  # noqa: F401
from .lib import LdError  # noqa: F401
from . import core, losses, resnet, fpn, heads, detectors  # noqa: F401,E402
from .registry import (build_backbone, build_detector, build_head,  # noqa
                       build_loss, build_neck)

__version__ = '0.1.0'
