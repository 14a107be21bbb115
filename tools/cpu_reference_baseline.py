"""BASELINE.md section 3 in the BUILD container (the only place the reference
exists): time the REFERENCE ITSELF -- KnowledgeDistillationSingleStageDetector
.forward_train -> _parse_losses -> backward -> torch.optim.SGD.step, imported
unchanged from /root/reference through oracle/ref_shim.py -- and the oracle
port (oracle/net_oracle.py, what bench.py's cpu_baseline times on the GPU box)
on the same machine, same synthetic batch and seeded weights: 1 warm-up +
median of 3, per-stage split, core count and CPU model stated.  The ratio
calibrates the port against the reference's CPU path.

    python tools/cpu_reference_baseline.py --out profiles/r02_cpu_reference_baseline.json
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'oracle'))


def median_run(fn, reps):
    fn()  # warm-up
    runs = []
    for _ in range(reps):
        tm = {}
        t0 = time.perf_counter()
        fn(tm)
        tm['total'] = time.perf_counter() - t0
        runs.append(tm)
    runs.sort(key=lambda r: r['total'])
    return runs[len(runs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(
        REPO, 'profiles', 'r02_cpu_reference_baseline.json'))
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--pad', default='800x1344')
    args = ap.parse_args()
    import gen_golden as G  # installs the shim, imports the reference
    import net_oracle as NO
    from ld_amd import synthetic
    hp, wp = (int(v) for v in args.pad.split('x'))
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    model = 'unknown'
    for line in open('/proc/cpuinfo'):
        if line.startswith('model name'):
            model = line.split(':', 1)[1].strip()
            break
    batch = synthetic.synthetic_batch(2, (hp, wp - 11 if wp == 1344 else wp),
                                      (hp, wp), 7, 1234)
    det = G.build_reference_detector(
        'configs/ld/ld_r50_gflv1_r101_fpn_coco_1x.py',
        imitation_method='finegrained')
    ssd = synthetic.seeded_state_dict(det.state_dict(), seed=1)
    tsd = synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2)
    det.load_state_dict(ssd)
    det.teacher_model.load_state_dict(tsd)
    det.train()
    opt = torch.optim.SGD([p for p in det.parameters() if p.requires_grad],
                          lr=0.0025, momentum=0.9, weight_decay=1e-4)

    def ref_step(tm=None):
        tm = tm if tm is not None else {}
        t0 = time.perf_counter()
        opt.zero_grad()
        # kd_one_stage.py:46-81, stage by stage
        x = det.extract_feat(batch['img'])
        t1 = time.perf_counter()
        with torch.no_grad():
            teacher_x = det.teacher_model.extract_feat(batch['img'])
            out_teacher = det.teacher_model.bbox_head(teacher_x)
        t2 = time.perf_counter()
        outs = det.bbox_head(x)
        t3 = time.perf_counter()
        losses = det.bbox_head.loss(*outs, batch['gt_bboxes'],
                                    batch['gt_labels'], out_teacher, x,
                                    teacher_x, batch['img_metas'])
        loss, _ = det._parse_losses(losses)
        t4 = time.perf_counter()
        loss.backward()
        t5 = time.perf_counter()
        opt.step()
        t6 = time.perf_counter()
        tm.update(student_net=(t1 - t0) + (t3 - t2), teacher_net=t2 - t1,
                  loss_block=t4 - t3, backward=t5 - t4, optimizer=t6 - t5)
        return float(loss)

    def port_step(tm=None):
        NO.ld_train_step(ssd, tsd, batch, 50, 101, with_backward=True,
                         timings=tm)

    ref = median_run(ref_step, args.reps)
    port = median_run(port_step, args.reps)
    n = batch['img'].shape[0]
    res = dict(
        machine=dict(cpu_model=model, threads=threads),
        workload=f'ld_r50_gflv1_r101_fpn_coco_1x, {n} x {hp}x{wp}, 7 GT/img, '
                 'fp32, seeded weights (ld_amd.synthetic)',
        method=f'1 warm-up + median of {args.reps} steps',
        reference=dict(kind='reference', images_per_s=n / ref['total'],
                       stages_s={k: round(v, 3) for k, v in ref.items()}),
        port=dict(kind='port', images_per_s=n / port['total'],
                  stages_s={k: round(v, 3) for k, v in port.items()}),
        port_over_reference=ref['total'] / port['total'])
    with open(args.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
