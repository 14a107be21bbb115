"""Vector-memory rate probe on the MI355X (tools/probe/probe.hip, built here
with hipcc): bytes per clock per CU for 4-byte and 16-byte per-lane loads from
buffers that fit L2 / the Infinity Cache / neither."""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libprobe.so')


def build():
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3',
                           '-std=c++17', '-shared', '-fPIC', '-o', SO,
                           os.path.join(HERE, 'probe.hip')])


def main():
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    dev = torch.device('cuda:0')
    res = []
    for mb in (2, 16, 128, 1024):
        nbytes = mb << 20
        buf = torch.randn(nbytes // 4, device=dev)
        for width in (1, 4):
            for blocks_per_cu in (1, 2, 4, 8):
                blocks = 256 * blocks_per_cu
                out = torch.empty(blocks * 256, device=dev)
                iters = 64
                st = torch.cuda.current_stream().cuda_stream

                def run():
                    rc = lib.probe_load_rate(C.c_void_p(buf.data_ptr()),
                                             C.c_uint(nbytes), width, blocks, iters,
                                             C.c_void_p(out.data_ptr()),
                                             C.c_void_p(st))
                    assert rc == 0
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(True), torch.cuda.Event(True)
                a.record()
                for _ in range(5):
                    run()
                b.record()
                torch.cuda.synchronize()
                dt = a.elapsed_time(b) * 1e-3 / 5
                moved = blocks * 4 * iters * 16 * 64 * 4 * width
                r = dict(buffer_mb=mb, bytes_per_lane=4 * width,
                         waves_per_simd=blocks_per_cu, tbps=moved / dt / 1e12,
                         bytes_per_clk_per_cu=moved / dt / 256 / 2.4e9)
                res.append(r)
                print(r, flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out',
                        'probe_load_rate.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(res, open(path, 'w'), indent=1)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        main()
