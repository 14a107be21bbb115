"""Which whole-step parameter gradients leave the tight element-wise band
(tests/_gradcheck.py) at the C2 size, under two DIFFERENT forward summation
orders (the shipped conv shape table vs one forced tile shape): a defect would
show in the same parameters both times, threshold flips (a ReLU input within
rounding noise of zero taking the other branch) move with the rounding.

    python tools/grad_outliers.py            (through gpurun)
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
from ld_amd import model_zoo, synthetic
class G:
    def __getitem__(self, n): return np.load(os.path.join(%r, 'tests', 'golden', n + '.npz'))
golden = G()
name = 'c2_r50'
g = golden['e2e']; cfg = g[name + '_cfg']
pad, img_shape, bseed = tuple(cfg[:2]), tuple(cfg[2:4]), int(cfg[4])
num_gt = [int(x) for x in g[name + '_num_gt']]
batch = synthetic.synthetic_batch(len(num_gt), img_shape, pad, num_gt, bseed)
dev = torch.device('cuda:0')
det = model_zoo.build_seeded_ld_detector(50, 101, dev, loss_im_weight=2.0)
d = dict(img=batch['img'].to(dev), img_metas=batch['img_metas'],
         gt_bboxes=[b.to(dev) for b in batch['gt_bboxes']],
         gt_labels=[l.to(dev) for l in batch['gt_labels']])
loss, _ = det._parse_losses(det(**d)); loss.backward(); torch.cuda.synchronize()
params = dict(det.named_parameters())
gs, gt = golden['grad_samples'], golden['grad_truth64']
rows = []
for k, ref, am, truth, referr in zip((str(x) for x in gs[name + '_grad_names']),
                                     gs[name + '_grad_samples'], gs[name + '_grad_absmax'],
                                     gt[name + '_grad_truth64'], gt[name + '_ref_abs_err']):
    flat = params[k].grad.reshape(-1)
    idx = synthetic.grad_sample_idx(flat.numel())
    got = flat[torch.from_numpy(idx).to(dev)].double().cpu().numpy()
    e_ref = float(np.abs(got - ref[:idx.size]).max() / am) if am else 0.0
    e_tru = float(np.abs(got - truth[:idx.size]).max() / am) if am else 0.0
    rows.append((e_tru, e_ref, float(referr) / am if am else 0.0, k))
rows.sort(reverse=True)
print('loss', float(loss))
print('vs float64: median ours %%.2e, median reference %%.2e' %% (
    np.median([r[0] for r in rows]), np.median([r[2] for r in rows])))
print('parameters with ours-vs-float64 > 3 x reference-vs-float64 + 5e-5 (of max|g|):')
for e_tru, e_ref, r_tru, k in rows:
    if e_tru > 3 * r_tru + 5e-5:
        print('  %%-44s ours %%.2e  reference %%.2e  (ours vs reference %%.2e)' %% (k, e_tru, r_tru, e_ref))
'''


def main():
    for tag, env in (('shipped shape table', {}),
                     ('forward / dgrad forced to the 2x2x2x8 tile', {'LD_CONV_STREAM': '2x2x2x8'}),
                     ('wave-private weight gradient', {'LD_CONV_WGRAD_CFG': '0'})):
        print('==', tag, flush=True)
        r = subprocess.run([sys.executable, '-c', CHILD % (REPO, REPO, REPO)],
                           env=dict(os.environ, **env), capture_output=True, text=True)
        print(r.stdout[-3000:] or r.stderr[-2000:], flush=True)


if __name__ == '__main__':
    main()
