"""ld_amd/csrc/ld_math.h (the scalar math of the gfx950 loss/target kernels)
compiled for the host and checked against the golden vectors / oracle.  This
does not exercise the GPU path -- it exists so formula errors are caught on the
CPU before a GPU session is spent."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import ld_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = C.c_float
FP = C.POINTER(C.c_float)


@pytest.fixture(scope='module')
def hh():
    out_dir = os.path.join(REPO, 'tests', '_build')
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, 'libhost_harness.so')
    src = os.path.join(REPO, 'tests', 'host_harness.cpp')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off',
                           '-shared', '-fPIC', src, '-o', so])
    lib = C.CDLL(so)
    for n in ('h_iou', 'h_diou', 'h_dist', 'h_giou', 'h_kl17', 'h_expect17',
              'h_dfl17', 'h_qfl_neg', 'h_qfl_pos', 'h_clamp_dist'):
        getattr(lib, n).restype = F
    lib.h_dist.argtypes = [F, F, F, F]
    lib.h_qfl_neg.argtypes = [F, FP]
    lib.h_qfl_pos.argtypes = [F, F, FP]
    lib.h_giou.argtypes = [FP, FP, F, FP, FP]
    lib.h_kl17.argtypes = [FP, FP, F, FP]
    lib.h_expect17.argtypes = [FP, FP]
    lib.h_dfl17.argtypes = [FP, F, FP, FP, C.POINTER(C.c_int)]
    lib.h_iou.argtypes = [FP, FP]
    lib.h_diou.argtypes = [FP, FP]
    lib.h_clamp_dist.argtypes = [F, F]
    return lib


def _p(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(FP)


def test_iou_family_bit_exact(hh, golden):
    g = golden['kat_losses']
    b1, b2 = g['giou_b1'], g['giou_b2']
    for i in range(3):
        for j in range(3):
            a, ap = _p(b1[i])
            b, bp = _p(b2[j])
            assert np.float32(hh.h_iou(ap, bp)) == g['kat6_pair_iou'][i, j]
            assert np.float32(hh.h_diou(ap, bp)) == g['kat6_pair_diou'][i, j]
    rng = np.random.RandomState(0)
    for _ in range(500):
        a = rng.rand(4).astype(np.float32) * 100
        a[2:] += a[:2]
        b = rng.rand(4).astype(np.float32) * 100
        b[2:] += b[:2]
        _, ap = _p(a)
        _, bp = _p(b)
        assert np.float32(hh.h_iou(ap, bp)) == O.bbox_overlaps(
            a[None], b[None])[0, 0]
        assert np.float32(hh.h_diou(ap, bp)) == O.bbox_overlaps(
            a[None], b[None], mode='diou')[0, 0]


def test_giou_kl_dfl_qfl_vs_golden(hh, golden):
    g = golden['kat_losses']
    # GIoU fwd + grad (KAT5)
    for i in range(3):
        _, pp = _p(g['giou_b1'][i])
        _, tp = _p(g['giou_b2'][i])
        iou, gr = C.c_float(), (C.c_float * 4)()
        l = hh.h_giou(pp, tp, 1e-6, C.byref(iou), gr)
        np.testing.assert_allclose(2 * l, g['kat5_none'][i], rtol=1e-6)
        np.testing.assert_allclose(iou.value, g['kat6_iou_aligned'][i],
                                   rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(
            np.array(list(gr)) * 2 * g['giou_w'][i], g['kat5_grad'][i],
            rtol=1e-4, atol=1e-7)
    # KL rows (KAT1, KAT1b)
    pred, soft, w = g['kl_pred'], g['kl_soft'], g['kl_w']
    for r in range(4):
        _, sp = _p(pred[r])
        _, tp = _p(soft[r])
        d = (C.c_float * 17)()
        kl = hh.h_kl17(sp, tp, 10.0, d)
        np.testing.assert_allclose(0.25 * kl, g['kat1_none'][r], rtol=2e-5)
        np.testing.assert_allclose(
            np.array(list(d)) * (10.0 / 17) * 0.25 * w[r] / 4,
            g['kat1_grad'][r], rtol=2e-4, atol=1e-9)
        # Integral (KAT3) and DFL (KAT2)
        p = (C.c_float * 17)()
        e = hh.h_expect17(sp, p)
        np.testing.assert_allclose(e, g['kat3_integral'][0, r], rtol=1e-6)
        wl, wr, yl = C.c_float(), C.c_float(), C.c_int()
        dl = hh.h_dfl17(sp, float(g['dfl_label'][r]), C.byref(wl),
                        C.byref(wr), C.byref(yl))
        np.testing.assert_allclose(0.25 * dl, g['kat2_none'][r], rtol=1e-5)
        gk = np.array(list(p))
        gk[yl.value] -= wl.value
        gk[yl.value + 1] -= wr.value
        np.testing.assert_allclose(gk * 0.25 * w[r] / 4, g['kat2_grad'][r],
                                   rtol=1e-4, atol=1e-7)
    # QFL (KAT4)
    x, labels, score = g['qfl_pred'], g['qfl_labels'], g['qfl_score']
    for r in range(6):
        tot = 0.0
        for c in range(5):
            dq = C.c_float()
            if labels[r] < 5 and c == labels[r]:
                q = hh.h_qfl_pos(float(x[r, c]), float(score[r]),
                                 C.byref(dq))
            else:
                q = hh.h_qfl_neg(float(x[r, c]), C.byref(dq))
            tot += q
            np.testing.assert_allclose(dq.value / 2.5, g['kat4_grad'][r, c],
                                       rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(tot, g['kat4_none'][r], rtol=1e-5)
