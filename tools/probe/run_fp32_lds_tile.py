"""fp32 1x1 conv with LDS-shared operands (tools/probe/fp32_lds_tile.hip) against
the shipped streaming / vector kernels of libldhip.so on the 1x1 layers of the
C2 step: correctness against a torch matmul, then time per launch.

    python tools/probe/run_fp32_lds_tile.py        (through gpurun)
"""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
SO = os.path.join(HERE, 'libfp32tile.so')

# (Cin, Cout, H, W) with N = 1 and H * W = the step's N * P columns
LAYERS = [(256, 1024, 100, 84), (1024, 256, 100, 84), (128, 512, 200, 168),
          (512, 128, 200, 168), (512, 2048, 50, 42), (2048, 512, 50, 42),
          (64, 256, 400, 336)]
SHAPES = [(1, 1, 16), (1, 1, 32), (2, 1, 16), (1, 2, 16), (2, 2, 16), (2, 1, 32),
          (1, 2, 32)]


def build():
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3',
                           '-std=c++17', '-shared', '-fPIC', '-o', SO,
                           os.path.join(HERE, 'fp32_lds_tile.hip')])


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def main():
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    from ld_amd import layers as Y
    dev = torch.device('cuda:0')
    torch.backends.cuda.matmul.allow_tf32 = False
    out = []
    for cin, cout, h, w in LAYERS:
        J = h * w
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(cin, J, generator=g).to(dev)
        wgt = (torch.randn(cout, cin, generator=g) / cin**0.5).to(dev)
        wt = wgt.t().contiguous()  # [K][Cout]
        ref = wgt.double() @ x.double()
        flops = 2.0 * cin * cout * J
        y = torch.empty(cout, J, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        row = dict(layer=f'{cin}>{cout} J{J}', gflop=flops / 1e9)
        for tm, tn, bk in SHAPES:
            if cin % bk or cout < 32 * tm:
                continue

            def run():
                rc = lib.probe_fp32_lds_tile(C.c_void_p(wt.data_ptr()),
                                             C.c_void_p(x.data_ptr()),
                                             C.c_void_p(y.data_ptr()), cin, cout, J,
                                             tm, tn, bk, C.c_void_p(st))
                assert rc == 0, rc
            y.fill_(float('nan'))
            run()
            torch.cuda.synchronize()
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            dt = timed(run)
            row[f'lds_{tm}x{tn}_bk{bk}'] = dict(us=dt * 1e6, tflops=flops / dt / 1e12,
                                               rel_err=err)
        # the shipped kernels on the same GEMM (N = 1 image of h x w positions)
        x3 = x.reshape(1, cin, J)
        w4 = wgt.reshape(cout, cin, 1, 1).contiguous()
        for tag, env in (('table_or_model', None), ('stream_1x1', '1x1x1x8x1'),
                         ('stream_2x1', '2x1x2x8x1'), ('stream_2x2', '2x2x2x8x1'),
                         ('vec_1x4', '1x4x1x8x2')):
            if env:
                os.environ['LD_CONV_STREAM'] = env
            else:
                os.environ.pop('LD_CONV_STREAM', None)
            try:
                yl, _ = Y.conv_forward_raw(x3, w4, 1, 0, ((h, w), ))
                torch.cuda.synchronize()
                err = float((yl[0].double() - ref).abs().max() / ref.abs().max())
                dt = timed(lambda: Y.conv_forward_raw(x3, w4, 1, 0, ((h, w), )))
                row['lib_' + tag] = dict(us=dt * 1e6, tflops=flops / dt / 1e12,
                                         rel_err=err)
            except Exception as e:  # a forced shape that does not fit
                row['lib_' + tag] = dict(error=str(e)[:80])
        os.environ.pop('LD_CONV_STREAM', None)
        best_l = max((v['tflops'], k) for k, v in row.items()
                     if k.startswith('lds_') and v["rel_err"] < 1e-4)
        best_s = max((v['tflops'], k) for k, v in row.items()
                     if k.startswith('lib_') and 'tflops' in v)
        row['best_lds'], row['best_lib'] = best_l, best_s
        print(row['layer'], 'LDS', f'{best_l[0]:.1f}', best_l[1], '| shipped',
              f'{best_s[0]:.1f}', best_s[1], flush=True)
        out.append(row)
    path = os.path.join(REPO, 'gpurun_out', 'probe_fp32_lds_tile.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
