"""CPU: sanity anchors of the DCNv1 oracle (oracle/dcn_oracle.py; parity
unpinned against mmcv, see its header): zero offsets reduce to an ordinary
convolution, integer offsets to a shifted one, and a hand-computed bilinear
sample."""
import torch
import torch.nn.functional as F


def test_zero_and_integer_offsets():
    import dcn_oracle as D
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 9, 11, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g)
    for stride in (1, 2):
        ref = F.conv2d(x, w, stride=stride, padding=1)
        off = torch.zeros(2, 18, ref.shape[2], ref.shape[3])
        torch.testing.assert_close(D.deform_conv2d(x, off, w, stride, 1), ref,
                                   rtol=1e-5, atol=1e-5)
    # every tap moved one row down and two columns right = conv of a shifted x
    off = torch.zeros(2, 18, 9, 11)
    off[:, 0::2] = 1.0
    off[:, 1::2] = 2.0
    shifted = torch.zeros_like(x)
    shifted[:, :, :-1, :-2] = x[:, :, 1:, 2:]
    # (rows / columns that touch the conv's zero padding differ by definition:
    # the deformed taps read real pixels there)
    torch.testing.assert_close(D.deform_conv2d(x, off, w, 1, 1)[:, :, 1:, 1:],
                               F.conv2d(shifted, w, padding=1)[:, :, 1:, 1:],
                               rtol=1e-5, atol=1e-5)


def test_bilinear_and_border_rule():
    import dcn_oracle as D
    x = torch.arange(12, dtype=torch.float32).view(1, 1, 3, 4)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0  # centre tap only: y(p) = sample at p + offset
    off = torch.zeros(1, 18, 3, 4)
    off[0, 8] = 0.5   # dy of tap 4
    off[0, 9] = 0.25  # dx of tap 4
    y = D.deform_conv2d(x, off, w, 1, 1)
    # at (0, 0): rows 0/1, cols 0/1 -> 0.5*(0.75*0 + 0.25*1) + 0.5*(0.75*4 + 0.25*5)
    assert abs(float(y[0, 0, 0, 0]) - 2.25) < 1e-6
    # last row: y = 2.5 -> row 3 is outside, contributes 0
    assert abs(float(y[0, 0, 2, 0]) - 0.5 * (0.75 * 8 + 0.25 * 9)) < 1e-6
    # beyond the border by >= 1: exactly 0
    off[0, 8] = 1.0
    y = D.deform_conv2d(x, off, w, 1, 1)
    assert float(y[0, 0, 2, 1]) == 0.0
