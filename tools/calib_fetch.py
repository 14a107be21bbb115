"""Known-byte streaming reads for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE
on gfx950 per access width (MI355X_MICROARCH.md: FETCH_SIZE reports half the
bytes of a 16 B/lane streaming read; other widths uncalibrated): a plain copy
of `n` floats with 4 B per lane (the fp32 conv's B-operand / epilogue pattern:
32 lanes = one 128-byte row segment) and with 16 B per lane.

    rocprofv3 --pmc FETCH_SIZE -- python tools/calib_fetch.py
Each launch reads 4 n bytes and writes 4 n bytes (n = 2^28: 1 GiB each way)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_amd import lib as L  # noqa: E402

dev = torch.device('cuda:0')
lib = L.get_lib()
n = 1 << 28
src = torch.randn(n, device=dev)
dst = torch.empty_like(src)
st = L.stream_ptr(dev)
for width in (4, 1):
    for _ in range(3):
        L.check(lib.ld_probe_copy(L.ptr(src), L.ptr(dst), n, width, 0, st), 'copy')
torch.cuda.synchronize()
print('bytes per launch each way:', 4 * n)
