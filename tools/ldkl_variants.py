"""A/B of the reg-side LD-KL kernel variants on the MI355X (run through gpurun).

  * correctness of every variant against the round-2 kernel (variant -1) on a
    C2-shaped batch WITH positives: loss table and reg gradient;
  * launch time of every variant at the saturating 2^24-row size (bench.py's
    roofline_ldkl geometry) and at the C2 step size;
  * HBM ceilings measured in the same process: plain copies and the kernel's own
    34-read / 17-write plane pattern with the arithmetic removed
    (ld_probe_copy / ld_probe_planes, ld_amd/csrc/probe.hip).

Writes gpurun_out/ldkl_variants.json.
"""
import ctypes as C
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from ld_amd import lib as L  # noqa: E402
from ld_amd import lossblock as LB  # noqa: E402
from ld_amd import synthetic  # noqa: E402

def variant_word(vec=1, ntl=1, nts=1, fast=0, side_fast=0, small_nt=0, w8=0,
                 lds_kb=0, xcd=0):
    return ({1: 0, 2: 1, 4: 2}[vec] | ntl << 2 | nts << 3 | fast << 4 |
            side_fast << 5 | small_nt << 6 | w8 << 7 | lds_kb << 8 | xcd << 16)


def describe(v):
    if v < 0:
        return 'round-2 kernel'
    return (f'vec{1 << (v & 3)} ntl{v >> 2 & 1} nts{v >> 3 & 1} '
            f'fast{v >> 4 & 1} sidefast{v >> 5 & 1} smallnt{v >> 6 & 1} '
            f'w8{v >> 7 & 1} lds{v >> 8 & 255}k xcd{v >> 16 & 1}')


def med_us(fn, warm, iters):
    us, us_min = bench._median_launch_us(fn, warm, iters)
    return us, us_min


def correctness(lib, dev, variants):
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    strides = [8, 16, 32, 64, 128]
    b = synthetic.synthetic_batch(2, (800, 1333), (800, 1344), 7, 99)
    hp = LB.make_hp()
    tg = LB.atss_targets(sizes, strides, b['img_metas'],
                         [x.to(dev) for x in b['gt_bboxes']],
                         [x.to(dev) for x in b['gt_labels']], hp, dev)
    g = torch.Generator(device='cpu').manual_seed(5)

    def rnd(c, scale):
        return [(torch.randn(2, c, h, w, generator=g) * scale).to(dev)
                for h, w in sizes]
    cls, reg, tcls, treg = rnd(80, 2.0), rnd(68, 3.0), rnd(80, 2.0), rnd(68, 3.0)
    x, tx = rnd(256, 1.0), rnd(256, 1.0)
    out = {}
    ref = None
    for v in [-1] + variants:
        lib.ld_loss_set_reg_variant(v)
        losses, grads, _, _ = LB.loss_block_forward(hp, tg, cls, reg, tcls, treg,
                                                    x, tx)
        torch.cuda.synchronize()
        tab = losses.double().cpu()
        gr = torch.cat([t.flatten() for t in grads['reg']]).double().cpu()
        if ref is None:
            ref = (tab, gr)
            continue
        dt = (tab - ref[0]).abs().max().item()
        rel_t = dt / max(ref[0].abs().max().item(), 1e-30)
        dg = (gr - ref[1]).abs().max().item()
        rel_g = dg / max(ref[1].abs().max().item(), 1e-30)
        out[v] = dict(loss_table_max_abs=dt, loss_table_rel=rel_t,
                      grad_max_abs=dg, grad_rel_to_max=rel_g,
                      ok=bool(rel_t < 2e-5 and rel_g < 2e-5))
    npos = int((tg['labels'] < 80).sum())
    return out, npos


def main():
    dev = torch.device('cuda:0')
    lib = L.get_lib()
    res = dict(device=torch.cuda.get_device_name(0))
    variants = []
    if os.environ.get('LDKL_ROUND', '2') == '1':  # the first sweep (session 3)
        for vec in (1, 2, 4):
            for ntl, nts in ((1, 1), (0, 1), (1, 0), (0, 0)):
                for fast in (0, 1):
                    variants.append(variant_word(vec, ntl, nts, fast))
        for fast in (0, 1):
            variants.append(variant_word(1, 1, 1, fast, w8=1))
            variants.append(variant_word(1, 1, 1, fast, side_fast=1))
            variants.append(variant_word(1, 1, 1, fast, side_fast=1, w8=1))
            variants.append(variant_word(2, 1, 1, fast, side_fast=1))
            variants.append(variant_word(4, 1, 1, fast, side_fast=1))
        for kb in (24, 40, 80):  # throttles: 6 / 4 / 2 workgroups per CU
            variants.append(variant_word(1, 1, 1, 1, lds_kb=kb))
            variants.append(variant_word(4, 1, 1, 1, lds_kb=kb))
    else:  # second sweep: around the side-fast winner
        variants.append(variant_word(1, 1, 1, 0))
        for fast in (0, 1):
            for vec in (1, 2):
                variants.append(variant_word(vec, 1, 1, fast, side_fast=1))
                variants.append(variant_word(vec, 1, 1, fast, side_fast=1, xcd=1))
        variants.append(variant_word(1, 0, 1, 1, side_fast=1))
        variants.append(variant_word(1, 1, 0, 1, side_fast=1))
        variants.append(variant_word(1, 1, 1, 1, side_fast=1, w8=1, xcd=1))
        for kb in (20, 24, 32, 40):  # 8 / 6 / 5 / 4 workgroups per CU
            variants.append(variant_word(1, 1, 1, 1, side_fast=1, lds_kb=kb))
            variants.append(variant_word(1, 1, 1, 1, side_fast=1, lds_kb=kb,
                                         xcd=1))
    corr, npos = correctness(lib, dev, variants)
    res['correctness_num_pos'] = npos
    res['correctness'] = {describe(v): r for v, r in corr.items()}
    bad = [describe(v) for v, r in corr.items() if not r['ok']]
    print('correctness: positives', npos, 'bad variants', bad, flush=True)

    # ---- ceilings -----------------------------------------------------------
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 29  # 2 GiB per array
    src = torch.randn(n, device=dev)
    dst = torch.empty_like(src)
    ceil = []
    for width in (1, 4):
        for nt in (0, 1):
            def run():
                L.check(lib.ld_probe_copy(L.ptr(src), L.ptr(dst), n, width, nt,
                                          C.c_void_p(st)), 'ld_probe_copy')
            us, usm = med_us(run, 3, 11)
            ceil.append(dict(kind='copy', bytes_per_lane=4 * width, nt=nt,
                             us=us, tbps=2 * 4 * n / us / 1e6,
                             tbps_best=2 * 4 * n / usm / 1e6))
            print(ceil[-1], flush=True)
    del src, dst
    rows = 1 << 22
    s = torch.randn(68 * rows, device=dev)
    t = torch.randn(68 * rows, device=dev)
    g = torch.empty_like(s)
    for nt in (0, 1):
        for sf in (0, 1):
            def run():
                L.check(lib.ld_probe_planes(L.ptr(s), L.ptr(t), L.ptr(g), rows,
                                            nt, sf, C.c_void_p(st)),
                        'ld_probe_planes')
            us, usm = med_us(run, 3, 21)
            ceil.append(dict(kind='planes 34r+17w, no math', nt=nt,
                             side_fast=sf, us=us,
                             tbps=204 * 4 * rows / us / 1e6,
                             tbps_best=204 * 4 * rows / usm / 1e6))
            print(ceil[-1], flush=True)
    del s, t, g
    res['ceilings'] = ceil
    torch.cuda.empty_cache()

    # ---- the kernel at the saturating size ------------------------------------
    launch, nrows = bench._reg_dense_launcher(dev, [(2048, 2048)], [8], 1, 1.0)
    big = []
    for rep in range(2):  # two interleaved passes: drift shows up as disagreement
        for v in [-1] + variants:
            lib.ld_loss_set_reg_variant(v)
            us, usm = med_us(launch, 3, 15)
            big.append(dict(variant=v, desc=describe(v), rep=rep, us=us,
                            us_min=usm, tbps=nrows * 207.0 / us / 1e6,
                            frac=nrows * 207.0 / us / 1e6 / 8.0))
            print(big[-1], flush=True)
    res['rows_2p24'] = big
    del launch
    torch.cuda.empty_cache()
    c2, rows_c2 = bench._reg_dense_launcher(
        dev, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)],
        [8, 16, 32, 64, 128], 2, 0.09, seed=2)
    small = []
    for v in [-1] + variants:
        for smallnt in (0, 1):
            if v < 0 and smallnt:
                continue
            vv = v | (smallnt << 6) if v >= 0 else v
            lib.ld_loss_set_reg_variant(vv)
            us, usm = med_us(c2, 5, 31)
            small.append(dict(variant=vv, desc=describe(vv), us=us, us_min=usm))
    res['c2_step_size'] = small
    best = sorted(small, key=lambda r: r['us'])[:6]
    print('C2 best', best, flush=True)
    lib.ld_loss_set_reg_variant(variant_word(1, 1, 1, 1, side_fast=1))
    out = os.path.join(REPO, 'gpurun_out', 'ldkl_variants_r%s.json' % os.environ.get('LDKL_ROUND', '2'))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    tops = sorted([r for r in big], key=lambda r: r['us'])[:10]
    print('TOP', json.dumps(tops, indent=1))


if __name__ == '__main__':
    main()
