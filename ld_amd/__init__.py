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
# queues fix that.  Round 4 found hipGraph replays collapsing with more than 4
# queues (bf16 15.2 -> 28 ms); round 5 found why (profiles/r05_graph_queues_s1.
# jsonl): a graph launch spreads the captured branches over
# DEBUG_HIP_FORCE_GRAPH_QUEUES internal streams (default 4), with 8 hardware
# queues each of them gets a queue of its own and every fork / join edge of the
# step (one pair per weight gradient) becomes a cross-queue dependency; held to 2
# graph streams -- the step has two lanes, DESIGN.md section 4 -- the replay is
# back at 15.2 ms with 8 queues.  Both variables are read when the runtime
# initialises (the first HIP call): set at import in a multi-process job, unless
# the application chose values itself; ld_amd.train warns when it came too late.
if int(_os.environ.get('WORLD_SIZE', '1') or 1) > 1 or \
        _os.environ.get('LD_FORCE_COLLECTIVES') == '1':
    _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    _os.environ.setdefault('DEBUG_HIP_FORCE_GRAPH_QUEUES', '2')

from . import registry  # noqa: F401,E402
from .config import Config, ConfigDict  # noqa: F401
from .lib import LdError  # noqa: F401
from . import core, losses, resnet, fpn, heads, detectors  # noqa: F401,E402
from .registry import (build_backbone, build_detector, build_head,  # noqa
                       build_loss, build_neck)

__version__ = '0.1.0'
