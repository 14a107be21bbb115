"""Pin down ds_read_b64_tr_b16 on the MI355X (tools/probe/tr_probe.hip): LDS slot
i holds the value i (16-bit elements); print what each lane receives for a few
per-lane address patterns."""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libtrprobe.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3',
                       '-std=c++17', '-shared', '-fPIC', '-o', SO,
                       os.path.join(HERE, 'tr_probe.hip')])
lib = C.CDLL(SO)
dev = torch.device('cuda:0')


def run(name, addr_of_lane):
    addr = torch.tensor([addr_of_lane(l) for l in range(64)], dtype=torch.int32,
                        device=dev)
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    rc = lib.tr_probe(C.c_void_p(addr.data_ptr()), C.c_void_p(out.data_ptr()),
                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype('uint16').reshape(64, 4)
    print('==', name)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 47, 48, 63):
        print('  lane %2d addr %5d (elem %4d) ->' % (l, addr_of_lane(l),
                                                    addr_of_lane(l) // 2),
              list(o[l]))
    return o


ROW = 64  # bytes per row in these patterns (32 elements)
# A: lane t of a 16-lane group addresses row t>>2, 8-byte chunk t&3; groups 1 KiB apart
run('A rows=t>>2 chunk=t&3 rowstride 64B, group stride 1024B',
    lambda l: (l >> 4) * 1024 + ((l & 15) >> 2) * ROW + (l & 3) * 8)
# B: every lane of a group passes the same base (uniform)
run('B uniform base per group', lambda l: (l >> 4) * 1024)
# C: lane t addresses row t&3, chunk t>>2
run('C rows=t&3 chunk=t>>2',
    lambda l: (l >> 4) * 1024 + ((l & 15) & 3) * ROW + ((l & 15) >> 2) * 8)
# D: linear 8 bytes per lane
run('D linear 8 B per lane', lambda l: l * 8)
