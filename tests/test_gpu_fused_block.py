"""GPU test (-m gpu) of the fused frozen-teacher bottleneck (csrc/conv_fused.hip,
VERDICT r4 next #1a): one launch must reproduce the three fused conv+BN(+ReLU)
launches of the C8-only trunk BIT FOR BIT -- same bf16 operands, same fp32
accumulation order, intermediates rounded to bf16 at the same places -- on the
R101 layer3 shape and on ragged / tiny maps (tiles overhanging the image, halo
rows and columns outside it)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _block(dev, seed):
    from ld_amd.resnet import Bottleneck
    torch.manual_seed(seed)
    blk = Bottleneck(1024, 256).to(dev).eval()
    with torch.no_grad():
        for m in blk.modules():
            if hasattr(m, 'running_mean'):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        for c in (blk.conv1, blk.conv2, blk.conv3):
            c.weight.normal_(0, (2.0 / (c.weight.shape[1] * c.weight.shape[2]**2))**0.5)
    for p in blk.parameters():
        p.requires_grad = False
    return blk


@pytest.mark.parametrize('shape', [(2, 50, 84), (1, 13, 21), (2, 7, 11), (1, 4, 16),
                                   (1, 5, 33), (1, 2, 2)])
def test_fused_bottleneck_equals_three_launches(shape):
    from ld_amd import layers as Y
    dev = torch.device('cuda:0')
    N, H, W = shape
    blk = _block(dev, 7 + H)
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(N, 1024, H * W, generator=g).to(dev).relu()
    prev = Y.get_precision()
    Y.set_precision('bf16')
    try:
        with torch.no_grad(), Y.c8_only_scope():
            x8 = Y.C8Act(Y.to_c8(x), x.shape)
            Y._FUSED_BLOCK[0] = False
            ref, _ = blk.forward3(x8, ((H, W), ))
            Y._FUSED_BLOCK[0] = True
            assert Y.fused_bottleneck_available(x8, 1024, 256, ((H, W), ))
            got, _ = blk.forward3(x8, ((H, W), ))
        torch.cuda.synchronize()
    finally:
        Y._FUSED_BLOCK[0] = True
        Y.set_precision(prev)
    assert isinstance(ref, Y.C8Act) and isinstance(got, Y.C8Act)
    a, b = ref.buf.view(torch.int16), got.buf.view(torch.int16)
    if not torch.equal(a, b):
        fa, fb = ref.float(), got.float()
        bad = (a != b)
        raise AssertionError(
            f'{int(bad.sum())} of {bad.numel()} bf16 words differ; max |diff| '
            f'{float((fa - fb).abs().max()):.4e}, scale {float(fa.abs().max()):.3e}')
    assert float(got.float().abs().max()) > 0.1
