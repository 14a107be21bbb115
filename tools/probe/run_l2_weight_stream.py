"""L2-resident weight-stream probe (tools/probe/l2_weight_stream.hip): every
workgroup streams the same 2.2 MB buffer (the bf16 weight set of one R101
layer3 bottleneck) with 16-byte loads, at 4 / 8 / 16 waves per CU and at the
grid sizes a fused bottleneck kernel would have (84 / 168 workgroups).  Reports
GB/s per CU and bytes per clock per CU (clock read from rocm-smi when
available, else the 2.1 GHz the conv kernels sustain).
    python tools/probe/run_l2_weight_stream.py [out.json]      (through gpurun)"""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libl2stream.so')


def build():
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3',
                           '-std=c++17', '-shared', '-fPIC', '-o', SO,
                           os.path.join(HERE, 'l2_weight_stream.hip')])


def main():
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    dev = torch.device('cuda:0')
    nbytes = 2 * (256 * 1024 + 256 * 2304 + 1024 * 256)  # 2.23 MB of bf16
    nbytes = nbytes // 4096 * 4096
    buf = torch.randint(0, 1 << 30, (nbytes // 4, ), dtype=torch.int32, device=dev)
    out = torch.empty(1024 * 4 * 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    clk_ghz = 2.1
    res = []
    for blocks in (84, 168, 256, 512, 1024):
        for unroll in (4, 8, 16):
            for skew in (0, 7):
                reps = 8

                def run():
                    rc = lib.probe_weight_stream(
                        C.c_void_p(buf.data_ptr()), C.c_uint(nbytes), blocks, reps,
                        unroll, C.c_uint(skew), C.c_void_p(out.data_ptr()),
                        C.c_void_p(st))
                    assert rc == 0
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(True), torch.cuda.Event(True)
                a.record()
                for _ in range(5):
                    run()
                b.record()
                torch.cuda.synchronize()
                dt = a.elapsed_time(b) * 1e-3 / 5
                cus = min(blocks, 256)
                per_wg = nbytes * reps / dt          # B/s one workgroup pulls
                per_cu = per_wg * blocks / cus
                r = dict(buffer_bytes=nbytes, workgroups=blocks,
                         waves_per_cu=4 * max(1, blocks // 256), loads_in_flight=unroll,
                         skew_rows=skew, us_per_pass_of_buffer=dt / reps * 1e6,
                         gbps_per_cu=per_cu / 1e9,
                         bytes_per_clk_per_cu=per_cu / (clk_ghz * 1e9),
                         aggregate_tbps=per_wg * blocks / 1e12)
                res.append(r)
                print(r, flush=True)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(HERE)), 'gpurun_out', 'probe_l2_weight_stream.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(dict(assumed_clock_ghz=clk_ghz, results=res), open(path, 'w'), indent=1)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        main()
