"""fp32 weight-gradient sweep on the MI355X (run through gpurun): every wgrad
geometry of the C2 train step x {wave-private kernel, workgroup-tiled kernel
instances (kg, bk) x split counts x fused / slab combination}.  Each candidate
is checked against the wave-private result, then event-timed.  One JSON record
per (shape, candidate); the per-shape winners go to stdout as a table and, with
--table, as shape-table records (MODE 2) that ld_conv_tune_load reads.

    python tools/wgrad_sweep.py --out gpurun_out/wgrad_sweep.json \
        [--table gpurun_out/wgrad_table.txt] [--quick]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

HEAD = ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))
# (name, launches per C2 step, N, Cin, Cout, k, stride, pad, input levels)
SHAPES = [
    ('head_tower', 8, 2, 256, 256, 3, 1, 1, HEAD),
    ('head_cls', 1, 2, 256, 80, 3, 1, 1, HEAD),
    ('head_reg', 1, 2, 256, 68, 3, 1, 1, HEAD),
    ('fpn_p3', 1, 2, 256, 256, 3, 1, 1, ((100, 168), )),
    ('l3_conv2', 6, 2, 256, 256, 3, 1, 1, ((50, 84), )),
    ('fpn_p5', 1, 2, 256, 256, 3, 1, 1, ((25, 42), )),
    ('fpn_p6', 1, 2, 256, 256, 3, 2, 1, ((25, 42), )),
    ('fpn_p7', 1, 2, 256, 256, 3, 2, 1, ((13, 21), )),
    ('l3_conv2_s2', 1, 2, 256, 256, 3, 2, 1, ((100, 168), )),
    ('l3_conv1', 6, 2, 1024, 256, 1, 1, 0, ((50, 84), )),
    ('l3_conv3', 6, 2, 256, 1024, 1, 1, 0, ((50, 84), )),
    ('l2_conv2', 3, 2, 128, 128, 3, 1, 1, ((100, 168), )),
    ('l2_conv2_s2', 1, 2, 128, 128, 3, 2, 1, ((200, 336), )),
    ('l2_conv3', 4, 2, 128, 512, 1, 1, 0, ((100, 168), )),
    ('l2_conv1', 3, 2, 512, 128, 1, 1, 0, ((100, 168), )),
    ('l2_conv1_first', 1, 2, 256, 128, 1, 1, 0, ((200, 336), )),
    ('l2_down', 1, 2, 256, 512, 1, 2, 0, ((200, 336), )),
    ('lat_c3', 2, 2, 512, 256, 1, 1, 0, ((100, 168), )),
    ('l3_down', 1, 2, 512, 1024, 1, 2, 0, ((100, 168), )),
    ('l4_conv2', 2, 2, 512, 512, 3, 1, 1, ((25, 42), )),
    ('l4_conv2_s2', 1, 2, 512, 512, 3, 2, 1, ((50, 84), )),
    ('l4_conv3', 3, 2, 512, 2048, 1, 1, 0, ((25, 42), )),
    ('l4_conv1', 2, 2, 2048, 512, 1, 1, 0, ((25, 42), )),
    ('l4_conv1_first', 1, 2, 1024, 512, 1, 1, 0, ((50, 84), )),
    ('l4_down', 1, 2, 1024, 2048, 1, 2, 0, ((50, 84), )),
    ('lat_c5', 1, 2, 2048, 256, 1, 1, 0, ((25, 42), )),
]
TILE_SHAPES = [(1, 32), (2, 32), (4, 32), (2, 64), (4, 64)]


def slots(kg, bk):
    lds = 2 * 2 * 128 * (bk + 2) * 4 + 256
    per_cu = min((160 * 1024) // lds, 2048 // (256 * kg))
    if per_cu * kg * 4 > 16:
        per_cu = 16 // (kg * 4)
    return 256 * max(per_cu, 1)


def candidates(cin, cout, k, J, quick, fused=False):
    out = [(0, 0, 0, 0, 0)]
    tiles = -(-cout // 128) * -(-cin // 128) * k * k
    for kg, bk in TILE_SHAPES:
        sl = slots(kg, bk)
        mx = max(1, min(256, J // (2 * bk)))
        base = sl / tiles
        mults = (1.0, ) if quick else (0.25, 0.5, 0.75, 1.0, 1.5, 2.0)
        seen = set()
        for m in mults:
            for sp in {int(base * m), -(-int(base * m * 1000) // 1000)}:
                sp = max(1, min(mx, sp))
                if sp in seen:
                    continue
                seen.add(sp)
                out.append((1, kg, bk, sp, 0))
                if sp > 1 and fused:
                    out.append((1, kg, bk, sp, 1))
        if 1 not in seen:
            out.append((1, kg, bk, 1, 0))
    if k == 3:  # the three kw taps of a kernel row per workgroup, one workgroup per CU
        tiles3 = -(-cout // 128) * -(-cin // 128) * 3
        mx = max(1, min(256, J // 64))
        seen = set()
        for m in ((1.0, ) if quick else (0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0)):
            for sp in {int(256 / tiles3 * m), -(-int(256 / tiles3 * m * 1000) // 1000)}:
                sp = max(1, min(mx, sp))
                if sp not in seen:
                    seen.add(sp)
                    out.append((2, 0, 0, sp, 0))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out',
                                                  'wgrad_sweep.json'))
    ap.add_argument('--table', default='')
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--fused', action='store_true', help='also time the in-launch combination')
    args = ap.parse_args()
    from ld_amd import layers as Y
    from ld_amd import lib as L
    dev = torch.device('cuda:0')
    lib = L.get_lib()
    st = L.stream_ptr(dev)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    recs, table = [], []
    total_old = total_new = 0.0
    for name, count, N, cin, cout, k, s, p, levels in SHAPES:
        if args.only and name not in args.only.split(','):
            continue
        d, _ = Y.conv_desc(N, cin, cout, k, k, s, p, levels)
        g = torch.Generator().manual_seed(cin * 7 + cout)
        x = torch.randn(N, cin, d.Pin, generator=g).to(dev)
        dy = torch.randn(N, cout, d.Pout, generator=g).to(dev)
        need = lib.ld_conv_tune_wgrad_workspace_bytes(C.byref(d))  # every forced plan fits
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        J = N * d.Pout
        flop = 2.0 * J * cin * cout * k * k

        def run(acc=0):
            L.check(lib.ld_conv_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw),
                                      acc, L.ptr(ws), ws.numel(), st), 'wgrad')
        ref = None
        best = None
        for cand in candidates(cin, cout, k, J, args.quick, args.fused):
            os.environ['LD_CONV_WGRAD_CFG'] = ','.join(str(v) for v in cand)
            dw.fill_(float('nan'))
            try:
                run()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                recs.append(dict(shape=name, cand=cand, error=str(e)))
                continue
            if ref is None:
                ref = dw.clone()
                err = 0.0
            else:
                err = float((dw - ref).abs().max() / ref.abs().max())
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            best_us = None
            for _ in range(2):
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.reps
                best_us = us if best_us is None else min(best_us, us)
            r = dict(shape=name, count=count, cin=cin, cout=cout, k=k, stride=s,
                     J=J, cand=cand, us=round(best_us, 2),
                     tflops=round(flop / best_us / 1e6, 1), err=err)
            recs.append(r)
            if err < 1e-5 and (best is None or best_us < best['us']):
                best = r
        old = [r for r in recs if r.get('shape') == name and
               r.get('cand') == (0, 0, 0, 0, 0)][0]
        total_old += old['us'] * count
        total_new += best['us'] * count
        bad = [r for r in recs if r.get('shape') == name and
               (r.get('err', 0) >= 1e-5 or 'error' in r)]
        print(f"{name:16s} x{count} old {old['us']:8.1f} us {old['tflops']:6.1f} TF"
              f" | best {best['cand']} {best['us']:8.1f} us {best['tflops']:6.1f} TF"
              f" | bad {len(bad)}", flush=True)
        key = [2, cin, cout, k, k, s, p, J, len(levels), levels[0][0],
               levels[0][1], 0, 0, 0, 0, 0, 0, 0]
        c = best['cand']
        table.append(' '.join(str(v) for v in key) +
                     f'  {c[0]} {c[1]} {c[2]} {c[3]} {c[4]} 0')
        del x, dy, ws, dw
    os.environ.pop('LD_CONV_WGRAD_CFG', None)
    print(f'step total: old {total_old / 1e3:.3f} ms -> best {total_new / 1e3:.3f} ms')
    with open(args.out, 'w') as f:
        json.dump(recs, f)
    if args.table:
        with open(args.table, 'w') as f:
            f.write('# fp32 wgrad picks (MODE 2): kind kg bk splits fused\n')
            f.write('\n'.join(table) + '\n')


if __name__ == '__main__':
    main()
