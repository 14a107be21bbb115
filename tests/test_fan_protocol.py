"""The fan-out gradient protocol of ld_amd.layers (fan_in / fan_take / fan_give)
on CPU: toy consumers with plain torch arithmetic stand in for the conv
data-gradient launches.  What is checked is the BOOK-KEEPING the GPU path relies
on: a deposit is handed to the next participant exactly once, the last
participant returns the total, reshape-views resolve to one fan, leaves and
non-participants keep going through autograd's own sum, and a deposit nobody
collects raises at the end of the backward pass."""
import pytest
import torch

from ld_amd import layers as Y


class Taker(torch.autograd.Function):
    """y = k * x; fuses whatever was deposited into its own 'kernel'."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k, ctx.fan = k, Y.fan_in(ctx, 0, x)
        return x * k

    @staticmethod
    def backward(ctx, dy):
        addend = Y.fan_take(ctx.fan)
        dx = dy * ctx.k
        if addend is not None:
            dx = dx + addend.view(dx.shape)  # the epilogue sum
        return Y.fan_give(ctx.fan, dx), None


class Giver(torch.autograd.Function):
    """y = k * x; cannot fuse (never calls fan_take)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k, ctx.fan = k, Y.fan_in(ctx, 0, x)
        return x * k

    @staticmethod
    def backward(ctx, dy):
        return Y.fan_give(ctx.fan, dy * ctx.k), None


@pytest.fixture(autouse=True)
def _fan_on():
    prev = Y._FAN_ON[0]
    Y._FAN_ON[0] = True
    yield
    Y._FAN_ON[0] = prev


def _leaf(*shape):
    return torch.randn(*shape, requires_grad=True)


def test_two_takers_one_total():
    a = _leaf(2, 3, 4)
    h = a * 1.0                      # non-leaf: the fan tensor
    before = Y.FAN_STATS['fused']
    y = Taker.apply(h, 2.0).sum() + Taker.apply(h, 3.0).sum()
    y.backward()
    assert torch.allclose(a.grad, torch.full_like(a, 5.0))
    assert Y.FAN_STATS['fused'] == before + 1
    assert not Y._FAN_OPEN and getattr(h, '_ld_stash', None) is None


def test_views_resolve_to_one_fan():
    a = _leaf(2, 3, 4)
    h = a * 1.0
    y = Taker.apply(h.view(2, 12), 2.0).sum() + \
        Taker.apply(h.view(2, 3, 2, 2).reshape(2, 3, 4), 3.0).sum() + \
        Taker.apply(h, 4.0).sum()
    y.backward()
    assert torch.allclose(a.grad, torch.full_like(a, 9.0))
    assert h._ld_fan == 0


def test_non_participant_goes_through_autograd():
    a = _leaf(5)
    h = a * 1.0
    y = Taker.apply(h, 2.0).sum() + (h * 7.0).sum() + Taker.apply(h, 3.0).sum()
    y.backward()
    assert torch.allclose(a.grad, torch.full_like(a, 12.0))


def test_leaf_is_left_to_autograd():
    a = _leaf(5)
    (Taker.apply(a, 2.0).sum() + Taker.apply(a, 3.0).sum()).backward()
    assert torch.allclose(a.grad, torch.full_like(a, 5.0))
    assert not hasattr(a, '_ld_fan')
    b = _leaf(5)
    y1, y2 = Taker.apply(b, 2.0), Taker.apply(b, 3.0)
    y1.sum().backward()              # a partial backward over a leaf is fine
    assert torch.allclose(b.grad, torch.full_like(b, 2.0))


def test_uncollected_deposit_raises():
    a = _leaf(5)
    h = a * 1.0
    y1, y2 = Taker.apply(h, 2.0), Taker.apply(h, 3.0)
    with pytest.raises(RuntimeError, match='deposited'):
        y1.sum().backward()          # y2's branch never runs: loud, not silent
    assert not Y._FAN_OPEN


def test_switch_off():
    Y._FAN_ON[0] = False
    a = _leaf(5)
    h = a * 1.0
    (Taker.apply(h, 2.0).sum() + Taker.apply(h, 3.0).sum()).backward()
    assert torch.allclose(a.grad, torch.full_like(a, 5.0))
    assert not hasattr(h, '_ld_fan')


def test_common_buffer_detection():
    levels = ((4, 6), (2, 3), (1, 2))
    from ld_amd import lossblock as LB
    like = [torch.empty(2, 5, h, w) for h, w in levels]
    views = LB.alloc_level_views(like)
    buf = Y._common_buffer(views, levels, 2, 5)
    assert buf is not None and buf.shape == (2, 5, 32)
    for v in views:
        v.fill_(float(v.shape[-1]))
    assert buf[0, 0, :24].eq(6).all() and buf[1, 4, 30:].eq(2).all()
    # separate tensors, a wrong order or a missing level are not a common buffer
    assert Y._common_buffer([v.clone() for v in views], levels, 2, 5) is None
    assert Y._common_buffer([views[1], views[0], views[2]],
                            (levels[1], levels[0], levels[2]), 2, 5) is None
    assert Y._common_buffer([views[0], None, views[2]], levels, 2, 5) is None


def test_split_levels_backward_is_zero_copy():
    """SplitLevelsFn hands the loss block's (N, C, P) gradient buffer on as the
    gradient of the level-concatenated tensor -- the same memory, untouched."""
    from ld_amd import lossblock as LB
    levels = ((4, 6), (2, 3), (1, 2))
    a = _leaf(2, 5, 32)
    x3 = a * 1.0
    outs = Y.split_levels(x3, levels)
    assert [tuple(o.shape) for o in outs] == [(2, 5, 4, 6), (2, 5, 2, 3), (2, 5, 1, 2)]
    assert all(o.data_ptr() == x3.data_ptr() + 4 * off
               for o, off in zip(outs, (0, 24, 30)))  # views, not copies
    grads = LB.alloc_level_views([o.detach() for o in outs])
    for i, g in enumerate(grads):
        g.copy_(torch.randn(g.shape))

    class ToyLoss(torch.autograd.Function):  # what LDLossBlock does with its maps
        @staticmethod
        def forward(ctx, *xs):
            return sum(x.sum() for x in xs)

        @staticmethod
        def backward(ctx, g):
            return tuple(grads)

    seen = {}
    x3.register_hook(lambda g: seen.update(ptr=g.data_ptr()))
    ToyLoss.apply(*outs).backward()
    buf = Y._common_buffer(grads, levels, 2, 5)
    assert seen['ptr'] == buf.data_ptr()
    assert torch.equal(a.grad, buf)
