"""CPU: the GFLv2 / LDv2 oracle (oracle/ld_oracle.py ld_loss_block(kd=...),
oracle/net_oracle.py quality_tail / ldv2_*) against the golden vectors the
REFERENCE produced for LDv2Head.loss and the whole LDv2 train step
(oracle/gen_golden.py gen_lossblock_v2 / gen_e2e_v2) -- pins the oracle that
the GPU tests of GFocalHead / LDv2Head check the HIP path against."""
import numpy as np
import pytest
import torch

from ld_amd import synthetic

V2_HP = dict(lw_im=2.0)


def _close(a, b, rtol, atol, what):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64),
                               np.asarray(b, dtype=np.float64), rtol=rtol,
                               atol=atol, err_msg=what)


def _head_sd():
    """The seeded parameters gen_lossblock_v2 gave the reference head: seeds
    derive from the KEY NAMES, so the four reg_conf keys are enough."""
    shapes = {'reg_conf.0.weight': (64, 20, 1, 1), 'reg_conf.0.bias': (64, ),
              'reg_conf.2.weight': (1, 64, 1, 1), 'reg_conf.2.bias': (1, )}
    ref = {k: torch.zeros(v) for k, v in shapes.items()}
    return synthetic.seeded_state_dict(ref, seed=5)


@pytest.mark.parametrize('name', ['v2_small', 'v2_small_crowd', 'v2_c2'])
def test_ldv2_lossblock_oracle_vs_reference(golden, name):
    import net_oracle as NO
    g = golden['lossblock_v2']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed, hseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4], cfg[5]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    sizes = synthetic.level_shapes(pad)
    hi = synthetic.synthetic_head_inputs(len(num_gt), sizes, seed=hseed,
                                         num_classes=81)
    out = NO.ldv2_loss_step(_head_sd(), hi, batch, V2_HP)
    _close(out['losses'], g[name + '_losses'], 2e-5, 2e-6, 'loss table')
    _close([float(q.double().abs().mean()) for q in out['quality']],
           g[name + '_quality_abs_mean'], 1e-5, 0, 'quality')
    for k in _head_sd():
        ref = g[f'{name}_gparam_{k}']
        _close(out['g_params'][k].numpy(), ref, 2e-4,
               2e-6 * float(np.abs(ref).max()) + 1e-9, 'grad ' + k)
    for key, got in (('cls', out['g_cls_feat']), ('reg', out['g_reg']),
                     ('x', out['g_x'])):
        _close([float(t.double().abs().sum()) for t in got],
               g[f'{name}_g{key}_abs_sum'], 1e-4, 1e-7, f'|g{key}|')
        for l, t in enumerate(got):
            full = f'{name}_g{key}_{l}'
            if full in g.files:
                ref = g[full]
                _close(t.numpy(), ref, 1e-4,
                       1e-6 * float(np.abs(ref).max()) + 1e-12, full)
            else:
                flat = t.numpy().reshape(-1)
                ref = g[full + '_sample']
                _close(flat[np.arange(0, flat.size, 1009)], ref, 1e-4,
                       1e-6 * float(np.abs(ref).max()) + 1e-12, full)


@pytest.mark.parametrize('name', ['v2_tiny_r50', 'v2_small_r50'])
def test_ldv2_net_oracle_vs_reference(golden, name):
    """Whole LDv2 step of the net oracle against the reference run from
    configs/ldv2/ld_r50_gflv2_r101_fpn_1x.py (imitation 'finegrained')."""
    import net_oracle as NO
    from ld_amd import build_detector, model_zoo
    g = golden['e2e_v2']
    cfg = [int(v) for v in g[name + '_cfg']]
    pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), cfg[4]
    num_gt = [int(v) for v in g[name + '_num_gt']]
    batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
    det = build_detector(model_zoo.ldv2_detector(50, 101))
    assert list(det.state_dict().keys()) == \
        [str(k) for k in g[name + '_student_keys']]
    assert list(det.teacher_model.state_dict().keys()) == \
        [str(k) for k in g[name + '_teacher_keys']]
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2)
    out = NO.ldv2_train_step(ssd, tsd, batch, 50, 101, V2_HP)
    _close(out['losses'], g[name + '_losses'], 1e-4, 1e-5, 'loss table')
    names = [str(k) for k in g[name + '_grad_names']]
    norms = g[name + '_grad_norms']
    got = np.array([float(out['grads'][k].double().norm()) for k in names])
    _close(got, norms, 2e-3, 1e-7, 'grad norms')
