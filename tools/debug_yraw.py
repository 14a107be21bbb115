"""Which rows of the raw second output / dgamma are wrong (round-3 epilogue debug)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ld_amd import layers as Y
from ld_amd import lib as L

dev = torch.device('cuda:0')
for (N, cin, cout, hw) in ((1, 1024, 256, (25, 42)), (2, 64, 256, (20, 28))):
    g = torch.Generator().manual_seed(cin + cout)
    P = hw[0] * hw[1]
    x = torch.randn(N, cin, P, generator=g).to(dev)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin**0.5).to(dev)
    scale = (torch.rand(cout, generator=g) + .5).to(dev)
    shift = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(N, cout, P, generator=g).to(dev)
    plain, _ = Y.conv_forward_raw(x, w, 1, 0, (hw, ))
    for shape in (None, '1x1x1x8x1', '2x2x2x8x1', '1x2x2x8x1', '1x1x1x8x4'):
        if shape:
            os.environ['LD_CONV_STREAM'] = shape
        else:
            os.environ.pop('LD_CONV_STREAM', None)
        raw = torch.full_like(plain, float('nan'))
        y, _ = Y.conv_forward_raw(x, w, 1, 0, (hw, ), scale=scale, shift=shift,
                                  residual=res, relu=True, y_raw=raw)
        torch.cuda.synchronize()
        want = torch.relu(plain * scale[None, :, None] + shift[None, :, None] + res)
        bad_y = ((y - want).abs() > 1e-4).flatten(0)
        bad_raw = ~(raw == plain)
        rows = bad_raw.any(dim=2).any(dim=0).nonzero().flatten().tolist()
        print(f'{cin}>{cout} P{P} shape={shape}: y bad {int(((y - want).abs() > 1e-4).sum())}, '
              f'raw bad {int(bad_raw.sum())} nan {int(torch.isnan(raw).sum())} rows {rows[:24]}')
    os.environ.pop('LD_CONV_STREAM', None)
    # BN backward parameter gradients
    dy = torch.randn(N, cout, P, generator=g).to(dev)
    mean = (torch.randn(cout, generator=g) * .1).to(dev)
    var = (torch.rand(cout, generator=g) + .5).to(dev)
    gamma = (torch.rand(cout, generator=g) + .5)
    gr = gamma.clone().requires_grad_(True)
    import torch.nn.functional as F
    xin = plain.detach().cpu()
    out = F.relu(F.batch_norm(xin.view(N, cout, *hw), mean.cpu(), var.cpu(), gr,
                              shift.cpu(), False, 0.0, 1e-5))
    out.backward(dy.cpu().view(N, cout, *hw))
    gd = gamma.to(dev).requires_grad_(True)
    z = Y.bn_act(plain.detach().requires_grad_(True), gd, shift, mean, var, 1e-5,
                 None, True) if hasattr(Y, 'bn_act') else None
    if z is not None:
        z = z[0] if isinstance(z, tuple) else z
        z.backward(dy)
        err = (gd.grad.cpu() - gr.grad).abs()
        print('  bn_act dgamma max err', float(err.max()), 'bad', (err > 1e-3 * float(gr.grad.abs().max())).nonzero().flatten().tolist()[:20])
