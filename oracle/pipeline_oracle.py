"""CPU restatement of the reference's image pipeline arithmetic -- TEST
INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product path never does).

What the reference runs per image (configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py:
66-77): Resize(keep_ratio) -> RandomFlip -> Normalize(to_rgb) -> Pad(32)
(mmdet/datasets/pipelines/transforms.py:203-233, 416-450, 497-509, 570-585),
which call mmcv.imrescale (cv2.resize, INTER_LINEAR), mmcv.imflip,
mmcv.imnormalize and mmcv.impad_to_multiple.  mmcv (pinned 1.2.4 .. 1.3 by
mmdet/__init__.py:16-17) and OpenCV are third-party dependencies absent from
/root/reference and from this image, so the pixel arithmetic below restates
their PUBLISHED algorithm:

  cv2.resize, INTER_LINEAR, 8-bit: for destination index d the source
  coordinate is f = (d + 0.5) * (src / dst) - 0.5 (fp32), s = floor(f),
  f -= s; s < 0 -> (0, f = 0); s >= src - 1 -> (src - 1, f = 0); the two
  coefficients are cvRound((1 - f) * 2048) and cvRound(f * 2048) (11-bit fixed
  point, INTER_RESIZE_COEF_BITS); horizontal pass in int32, vertical pass
  (b0 * r0 + b1 * r1 + 2^21) >> 22, saturated to uint8.

PARITY UNPINNED for the resize (no cv2 here to generate vectors; OpenCV's SIMD
vertical pass is documented to differ from the scalar formula by at most 1 LSB
in rare cases).  The size rule, box transforms and the samplers ARE pinned
against the reference (tests/golden/pipeline.npz, oracle/gen_golden.py).
"""
import numpy as np


def _coef(dst, src):
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * (float(src) / float(dst)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    s1 = np.minimum(s + 1, src - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, a0, a1


def resize_linear_u8(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h), interpolation=INTER_LINEAR), uint8 HWC."""
    h, w = img.shape[:2]
    sx0, sx1, ax0, ax1 = _coef(new_w, w)
    sy0, sy1, ay0, ay1 = _coef(new_h, h)
    src = img.astype(np.int64)
    rows = (src[:, sx0] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None])
    out = (rows[sy0] * ay0[:, None, None] + rows[sy1] * ay1[:, None, None] +
           (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess(img, new_h, new_w, flip, mean, std, to_rgb, pad_h, pad_w):
    """One image through resize -> flip -> normalize -> pad; returns
    (3, pad_h, pad_w) fp32 (DefaultFormatBundle's HWC -> CHW transpose)."""
    r = resize_linear_u8(img, new_h, new_w)
    if flip:
        r = r[:, ::-1]  # mmcv.imflip 'horizontal' = np.flip(img, axis=1)
    x = r.astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]  # cv2.cvtColor(BGR2RGB)
    mean = np.asarray(mean, np.float32)
    stdinv = (1.0 / np.asarray(std, np.float32).astype(np.float64)).astype(
        np.float32)
    x = (x - mean) * stdinv  # cv2.subtract / cv2.multiply on float32
    out = np.zeros((3, pad_h, pad_w), np.float32)
    out[:, :new_h, :new_w] = x.transpose(2, 0, 1)
    return out
