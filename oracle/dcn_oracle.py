"""TEST INFRASTRUCTURE ONLY -- independent torch-CPU restatement of deformable
convolution v1 (forward), the checker of ld_amd/csrc/dcn.hip.  Never imported
by `ld_amd`.

PARITY UNPINNED: the op lives in mmcv-full (mmcv.ops.DeformConv2dPack, pinned
>=1.2.4,<=1.3 by mmdet/__init__.py:18-26; compiled CUDA, absent from
/root/reference), and the reference holds no golden vectors for it.  What is
restated is the published algorithm (Dai et al., "Deformable Convolutional
Networks", ICCV 2017, eq. 2-4) with mmcv's conventions
(deform_conv_cuda_kernel.cuh, deformable_im2col / _bilinear):
  * offset channels: (dy, dx) interleaved per tap k = kh*KW + kw;
  * sample at p = (ho*stride - pad + kh*dil + dy, wo*stride - pad + kw*dil + dx);
  * value 0 unless -1 < p_y < H and -1 < p_x < W; the four bilinear neighbours
    that fall outside the map contribute 0.
Call site anchored on: mmdet/models/backbones/resnet.py:171-194 and
configs/gfl/gfl_r101_fpn_dconv_c3-c5_mstrain_2x_coco.py:11-12 (Q9: DCN *v1*).
Written with dense tensor ops (one gather per bilinear corner), i.e. not the
thread-per-(tap, position) loop of the HIP kernel.
"""
import torch
import torch.nn.functional as F


def deform_sample(x, offset, k, stride, pad, dil=1):
    """x (N, C, H, W), offset (N, 2*k*k, Ho, Wo) -> (N, C, k*k, Ho, Wo)."""
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[2:]
    dev, dt = x.device, torch.float32
    ho = torch.arange(Ho, device=dev, dtype=dt).view(1, 1, Ho, 1)
    wo = torch.arange(Wo, device=dev, dtype=dt).view(1, 1, 1, Wo)
    kh = torch.arange(k, device=dev, dtype=dt).repeat_interleave(k).view(1, -1, 1, 1)
    kw = torch.arange(k, device=dev, dtype=dt).repeat(k).view(1, -1, 1, 1)
    off = offset.view(N, k * k, 2, Ho, Wo)
    py = ho * stride - pad + kh * dil + off[:, :, 0]
    px = wo * stride - pad + kw * dil + off[:, :, 1]
    inside = (py > -1) & (px > -1) & (py < H) & (px < W)
    y0, x0 = torch.floor(py), torch.floor(px)
    ly, lx = py - y0, px - x0
    flat = x.reshape(N, C, H * W)
    out = torch.zeros((N, C, k * k, Ho, Wo), dtype=dt, device=dev)
    for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx),
                        (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
        yy, xx = y0 + dy, x0 + dx
        ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long()
        g = torch.gather(flat.unsqueeze(2).expand(N, C, k * k, H * W), 3,
                         idx.view(N, 1, k * k, Ho * Wo).expand(N, C, k * k,
                                                               Ho * Wo))
        out += g.view(N, C, k * k, Ho, Wo) * (wgt * ok).unsqueeze(1)
    return out


def deform_conv2d(x, offset, weight, stride=1, pad=1, dil=1):
    """y = sum_{ci,k} W[co, ci, k] * sample[ci, k]  (deform_groups = 1)."""
    k = weight.shape[2]
    cols = deform_sample(x, offset, k, stride, pad, dil)
    N, C, K2, Ho, Wo = cols.shape
    return torch.einsum('ok,nkp->nop', weight.reshape(weight.shape[0], -1),
                        cols.reshape(N, C * K2, Ho * Wo)).view(N, -1, Ho, Wo)


def dcn_pack_forward(x, weight, off_w, off_b, stride=1, pad=1):
    """DeformConv2dPack.forward: offsets from the layer's own 3x3 conv."""
    offset = F.conv2d(x, off_w, off_b, stride=stride, padding=pad)
    return deform_conv2d(x, offset, weight, stride, pad), offset
