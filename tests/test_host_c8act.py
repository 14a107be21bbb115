"""Host-side logic of the C8-only activation carrier (ld_amd.layers.C8Act,
DESIGN.md section 3.4): pure bookkeeping, runs without a GPU."""
import pytest
import torch

from ld_amd import layers as Y
from ld_amd.lib import LdError


def _c8_image(x):
    """(N, C, P) fp32 -> the bf16 (N, C/8, P, 8) image, flattened: the layout
    ld_conv_to_c8 writes (tests/test_gpu_bf16.py::test_to_c8_layout)."""
    n, c, p = x.shape
    return x.to(torch.bfloat16).reshape(n, c // 8, 8, p).permute(
        0, 1, 3, 2).reshape(-1).contiguous()


def test_c8act_views_share_the_buffer_and_round_trip():
    x = torch.randn(2, 32, 6 * 5)
    a = Y.C8Act(_c8_image(x), x.shape)
    assert a.shape == (2, 32, 30) and a.dim() == 3 and a.size(1) == 32
    assert not a.requires_grad and a.dtype == torch.float32
    v = a.view(2, 32, 6, 5)
    assert v.buf is a.buf and v.shape == (2, 32, 6, 5)
    assert v.reshape(2, 32, -1).shape == (2, 32, 30)
    assert v.reshape((2, 32, 30)).shape == (2, 32, 30)
    want = x.to(torch.bfloat16).float()
    assert torch.equal(a.float(), want)
    assert torch.equal(v.float(), want.reshape(2, 32, 6, 5))


def test_c8act_refuses_views_that_touch_batch_or_channels():
    a = Y.C8Act(torch.zeros(2 * 32 * 30, dtype=torch.bfloat16), (2, 32, 30))
    for bad in ((2, 16, 60), (1, 64, 30), (4, 32, 15), (2, 32, 31), (2, 960)):
        with pytest.raises(LdError):
            a.reshape(*bad)


def test_c8_only_is_a_bf16_mode_feature():
    """The C8-only teacher trunk needs bf16 mode, the C8 path and channel counts
    that are multiples of 32; LD_TEACHER_C8_ONLY=0 switches it off."""
    prev = Y._PRECISION[0]
    try:
        Y._PRECISION[0] = 'fp32'
        assert not Y.c8_only_available([64, 256])
        Y._PRECISION[0] = 'bf16'
        assert Y.c8_only_available([64, 256, 2048])
        assert not Y.c8_only_available([64, 80])
        Y.set_c8(False)
        assert not Y.c8_only_available([64])
        Y.set_c8(True)
    finally:
        Y._PRECISION[0] = prev
        Y.set_c8(True)


def test_c8_only_scope_nests_and_restores():
    assert not Y._C8_ONLY[0]
    with Y.c8_only_scope():
        assert Y._C8_ONLY[0]
        with Y.c8_only_scope():
            assert Y._C8_ONLY[0]
        assert Y._C8_ONLY[0]
    assert not Y._C8_ONLY[0]


def test_resnet_keeps_fp32_activations_outside_the_conditions():
    """ResNet._c8_only: only with the flag, under no_grad, in eval mode, in
    bf16 mode (no GPU needed to evaluate the predicate)."""
    from ld_amd import build_backbone
    bb = build_backbone(dict(
        type='ResNet', depth=18, num_stages=4, out_indices=(0, 1, 2, 3),
        frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True),
        norm_eval=True, style='pytorch'))
    assert bb.train() is bb and bb.eval() is bb
    prev = Y._PRECISION[0]
    try:
        Y._PRECISION[0] = 'bf16'
        with torch.no_grad():
            assert not bb._c8_only()          # flag not set
            bb.c8_activations = True
            assert bb._c8_only()
            bb.train()
            assert not bb._c8_only()          # training mode
            bb.eval()
        assert not bb._c8_only()              # gradients enabled
        Y._PRECISION[0] = 'fp32'
        with torch.no_grad():
            assert not bb._c8_only()          # fp32 mode
    finally:
        Y._PRECISION[0] = prev


def test_retina_pseudo_image_views_share_memory():
    """RetinaGFLHead._pseudo: an (N, B * C, H, W) level map -- a strided slice
    of the level-concatenated (N, B * C, P) head output -- viewed as (N * B, C,
    H, W) WITHOUT a copy; gradients written through the view land in the
    original layout (what the fused loss block relies on)."""
    from ld_amd import model_zoo
    from ld_amd.registry import build_head
    head = build_head(dict(model_zoo.retina_gfl_detector(50)['bbox_head']))
    assert head.num_anchors == 9
    N, P = 2, 6 * 5 + 3 * 3
    full = torch.arange(N * 720 * P, dtype=torch.float32).reshape(N, 720, P)
    lv0 = full[:, :, :30].view(N, 720, 6, 5)      # non-contiguous level slice
    lv1 = full[:, :, 30:].view(N, 720, 3, 3)
    p0, p1 = head._pseudo([lv0, lv1], 80)
    assert p0.shape == (N * 9, 80, 6, 5) and p1.shape == (N * 9, 80, 3, 3)
    assert p0.data_ptr() == lv0.data_ptr() and p1.data_ptr() == lv1.data_ptr()
    # pseudo-image n * 9 + b, class c  ==  image n, channel b * 80 + c
    for n, b, c in ((0, 0, 0), (1, 4, 17), (1, 8, 79)):
        assert torch.equal(p0[n * 9 + b, c], lv0[n, b * 80 + c])
        assert torch.equal(p1[n * 9 + b, c], lv1[n, b * 80 + c])
    assert p0.stride() == (80 * P, P, 5, 1)
    # a contiguous gradient in the pseudo layout IS the (N, 720, H, W) gradient
    g = torch.randn(N * 9, 80, 6, 5)
    assert torch.equal(g.view(N, 720, 6, 5)[1, 4 * 80 + 17], g[1 * 9 + 4, 17])


def test_student_frozen_stages_run_c8_only_decision():
    """bf16 mode: a training student's frozen leading stages (frozen_stages = 1)
    run C8-only like the teacher's trunk; never in fp32 mode, under no_grad
    (inference keeps fp32 masters), or for the teacher-flagged backbone (its
    whole trunk already is)."""
    import torch
    from ld_amd import layers as Y
    from ld_amd import model_zoo
    from ld_amd.registry import build_backbone
    net = build_backbone(model_zoo._backbone(50)).train()
    assert net._frozen_c8_stages() == 0  # fp32 mode
    Y.set_precision('bf16')
    try:
        assert net._frozen_c8_stages() == 1
        with torch.no_grad():
            assert net._frozen_c8_stages() == 0
        net.c8_activations = True
        assert net._frozen_c8_stages() == 0
        net.c8_activations = False
        r18 = build_backbone(model_zoo._backbone(18)).train()
        # 64-channel BasicBlock stages: multiples of 32 -> eligible as well
        assert r18._frozen_c8_stages() == 1
        unfrozen = build_backbone(dict(model_zoo._backbone(50),
                                       frozen_stages=-1)).train()
        assert unfrozen._frozen_c8_stages() == 0
    finally:
        Y.set_precision('fp32')
