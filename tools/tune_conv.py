"""Explicit conv shape tuning on the MI355X (run through gpurun): one LD train
step of the benchmark config per precision mode with ld_amd.layers.autotune on
-- every conv geometry of the step calls ld_conv_tune_* / ld_conv_bf16_tune_*
once -- then the library's table is written out.  The result is what gets
committed as ld_amd/tune/gfx950.txt.

    python tools/tune_conv.py --out gpurun_out/tune_r02.txt [--modes fp32,bf16]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out',
                                                  'tune_table.txt'))
    ap.add_argument('--modes', default='fp32,bf16')
    ap.add_argument('--student', type=int, default=50)
    ap.add_argument('--fresh', action='store_true',
                    help='forget the shipped table first (retune everything)')
    ap.add_argument('--fresh-family', type=int, default=-1,
                    help='forget only the records of this kernel family (0 = '
                    'fp32 streaming) and retune those')
    args = ap.parse_args()
    os.environ.setdefault('LD_CONV_TUNE_LOG', '1')
    from ld_amd import layers as Y
    from ld_amd import lib as L
    from ld_amd import model_zoo, synthetic
    from ld_amd.train import SGDTrainer
    dev = torch.device('cuda:0')
    det = model_zoo.build_seeded_ld_detector(args.student, 101, dev)
    tr = SGDTrainer(det, lr=0.0025)
    b = synthetic.synthetic_batch(2, (800, 1333), (800, 1344), 7, 1234)
    d = dict(img=b['img'].to(dev), img_metas=b['img_metas'],
             gt_bboxes=[x.to(dev) for x in b['gt_bboxes']],
             gt_labels=[x.to(dev) for x in b['gt_labels']])
    Y.autotune(True)
    if args.fresh:
        L.get_lib().ld_conv_tune_clear()
    elif args.fresh_family >= 0:
        # keep the other families' records: dump, filter, reload
        tmp = args.out + '.keep'
        os.makedirs(os.path.dirname(tmp), exist_ok=True)
        L.save_tune_table(tmp)
        keep = [ln for ln in open(tmp)
                if ln.startswith('#') or not ln.strip() or
                int(ln.split()[16]) != args.fresh_family]
        with open(tmp, 'w') as f:
            f.writelines(keep)
        L.get_lib().ld_conv_tune_clear()
        L.get_lib().ld_conv_tune_load(tmp.encode())
        os.remove(tmp)
    # the teacher must not run concurrently while candidates are being timed
    det.use_teacher_stream = False
    for mode in args.modes.split(','):
        Y.set_precision(mode)
        # bf16: once with the C8 operand policy (family 2 keys for the layers
        # that use it), once without (family 1 keys for every layer)
        for c8 in ((True, False) if mode == 'bf16' else (True, )):
            Y.set_c8(c8)
            Y._TUNED.clear()
            tr.step(d)
            torch.cuda.synchronize()
        Y.set_c8(True)
    n = L.save_tune_table(args.out)
    print(f'wrote {n} records to {args.out}')


if __name__ == '__main__':
    main()
