"""Element-wise checks of whole-step parameter gradients (round 4).

Fixtures:
  grad_samples.npz  -- the REFERENCE's gradients, 256 sampled elements of every
      trainable parameter (oracle/gen_golden.py gen_grad_samples executes the
      reference; indices = synthetic.grad_sample_idx);
  grad_truth64.npz  -- the same elements with the nets evaluated in float64
      (gen_grad_truth64), and the reference's own deviation from them.

What fp32 can be held to: the reference's fp32 CPU gradients differ from the
float64 evaluation by up to 4e-4 of max|g| (median 2-4e-5; long cancelling sums
through ~100 conv layers), so two correct fp32 implementations differ by that
much.  On top of it come DISCRETE events: a ReLU input (or a clamp / max in the
loss block) within rounding noise of its threshold takes a different branch in
two implementations, and the gradients of the few parameters that element
feeds move by one element's worth (seen in round 4 as channel 143 of
backbone.layer3.3.bn1 in BOTH the GFL and the GFLv2 step -- same backbone
activations -- and in the reference itself: its bbox_head.reg_convs.1.gn.bias
is 3.6e-4 of max|g| off the float64 value, 10x its typical error).  The checks
therefore bound the BULK tightly and the few outliers loosely:
  (1) against the reference, per element:
          |got - ref| <= RTOL |ref| + ATOL_REL max|g|      (1e-3, 1e-3)
      for all but at most MAX_OUTLIERS parameters, and OUTLIER_REL max|g| for
      those -- element-wise: a dropped term, a sign or layout error in any
      sampled element shows.  Round 5 sized both constants to what was MEASURED
      (profiles/r04_grad_outliers_c2.txt: no parameter left this band under
      either forward summation order; the worst element 6.3e-4 of max|g|):
      OUTLIER_REL 1e-2 -> 2e-3, MAX_OUTLIERS 8 -> 5;
  (2) against float64, per parameter: our worst sampled error is at most
          3 x the reference's own worst error + 5e-5 max|g|
      for all but MAX_OUTLIERS_T64 parameters (measured: 5 under both
      summation orders tried, a different five each time), and our MEDIAN error over the
      parameters is within 2x the reference's median: we are as close to the
      exact gradient as the reference is."""
import numpy as np

RTOL, ATOL_REL = 1e-3, 1e-3
MAX_OUTLIERS = 5      # parameters (of ~175) allowed outside the tight band (1)
MAX_OUTLIERS_T64 = 6  # ... outside band (2): 5 measured, which five moves with the rounding
OUTLIER_REL = 2e-3    # ... but never further than this fraction of max|g| (worst seen 6.3e-4)


def _sampled(p, synthetic):
    import torch
    flat = p.grad.reshape(-1)
    idx = synthetic.grad_sample_idx(flat.numel())
    return idx, flat[torch.from_numpy(idx).to(flat.device)].double().cpu().numpy()


def check_grad_samples(golden, name, params, rtol=RTOL, atol_rel=ATOL_REL):
    from ld_amd import synthetic
    g = golden['grad_samples']
    names = [str(k) for k in g[name + '_grad_names']]
    bad, worst = [], []
    for k, ref, am in zip(names, g[name + '_grad_samples'], g[name + '_grad_absmax']):
        assert params[k].grad is not None, k
        idx, got = _sampled(params[k], synthetic)
        ref = ref[:idx.size].astype(np.float64)
        err = np.abs(got - ref)
        tol = rtol * np.abs(ref) + atol_rel * am + 1e-30
        ratio = float((err / tol).max())
        worst.append((float(err.max() / max(am, 1e-30)), k))
        if ratio > 1.0:
            i = int((err / tol).argmax())
            bad.append((k, int(idx[i]), float(got[i]), float(ref[i]), float(am),
                        round(ratio, 2)))
    worst.sort(reverse=True)
    assert len(bad) <= MAX_OUTLIERS, (
        f'{name}: {len(bad)} of {len(names)} parameter gradients off element-wise '
        f'vs the reference (rtol {rtol}, atol {atol_rel} * max|g|; at most '
        f'{MAX_OUTLIERS} threshold-flip outliers allowed); (name, flat index, got, '
        f'ref, max|g|, err/tol): {sorted(bad, key=lambda b: -b[5])[:6]}')
    assert worst[0][0] <= OUTLIER_REL, (
        f'{name}: {worst[0][1]} is {worst[0][0]:.2e} of max|g| off the reference')
    return worst  # [(max|err| / max|g|, name)], largest first


def check_grad_truth64(golden, name, params, factor=3.0, floor_rel=5e-5):
    from ld_amd import synthetic
    g, gs = golden['grad_truth64'], golden['grad_samples']
    names = [str(k) for k in g[name + '_grad_names']]
    absmax = dict(zip((str(k) for k in gs[name + '_grad_names']),
                      gs[name + '_grad_absmax']))
    bad, rows = [], []
    for k, truth, referr in zip(names, g[name + '_grad_truth64'],
                                g[name + '_ref_abs_err']):
        idx, got = _sampled(params[k], synthetic)
        err = float(np.abs(got - truth[:idx.size]).max())
        am = float(absmax[k])
        rows.append((err / max(am, 1e-30), float(referr) / max(am, 1e-30), k))
        if err > factor * float(referr) + floor_rel * am + 1e-30:
            bad.append((k, err, float(referr), am))
    rows.sort(reverse=True)
    assert len(bad) <= MAX_OUTLIERS_T64, (
        f'{name}: {len(bad)} of {len(names)} parameter gradients are further from '
        f'the float64 evaluation than {factor} x the reference itself (+ {floor_rel} '
        f'max|g|; at most {MAX_OUTLIERS_T64} outliers allowed); (name, our max err, '
        f'reference max err, max|g|): {bad[:6]}')
    ours = float(np.median([a for a, _, _ in rows]))
    theirs = float(np.median([b for _, b, _ in rows]))
    assert ours <= 2.0 * theirs + 1e-6, (
        f'{name}: median error vs float64 {ours:.2e} of max|g|, the reference\'s '
        f'{theirs:.2e}')
    assert rows[0][0] <= OUTLIER_REL
    return rows  # [(our err / max|g|, reference err / max|g|, name)]
