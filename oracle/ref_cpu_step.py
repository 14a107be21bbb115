"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ld_amd`.

Times the REFERENCE'S OWN LD train step on the host cores:
KnowledgeDistillationSingleStageDetector.forward_train -> _parse_losses ->
backward -> torch.optim.SGD.step (mmdet/models/detectors/kd_one_stage.py:46-81,
base.py:185-253), imported unmodified from LD_REFERENCE_ROOT through
oracle/ref_shim.py, on the synthetic C2 batch and seeded weights bench.py uses.
Run as a child process of bench.py's `cpu_baseline` leg (the shim rewires
sys.meta_path; the bench process stays clean).  Prints one JSON line.

    LD_REFERENCE_ROOT=<root> python oracle/ref_cpu_step.py --threads 64 --reps 3
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--pad', default='800x1344')
    ap.add_argument('--num-gt', type=int, default=7)
    ap.add_argument('--seed', type=int, default=1234)
    args = ap.parse_args()
    if args.reps < 1 or args.reps % 2 == 0:
        raise SystemExit('--reps must be odd: the reported step is the median')
    import torch
    torch.set_num_threads(args.threads)
    import gen_golden as G  # installs the shim, imports the reference
    from ld_amd import synthetic
    hp, wp = (int(v) for v in args.pad.split('x'))
    batch = synthetic.synthetic_batch(2, (hp, wp - 11 if wp == 1344 else wp),
                                      (hp, wp), args.num_gt, args.seed)
    det = G.build_reference_detector(
        'configs/ld/ld_r50_gflv1_r101_fpn_coco_1x.py',
        imitation_method='finegrained')
    det.load_state_dict(synthetic.seeded_state_dict(det.state_dict(), seed=1))
    det.teacher_model.load_state_dict(
        synthetic.seeded_state_dict(det.teacher_model.state_dict(), seed=2))
    det.train()
    opt = torch.optim.SGD([p for p in det.parameters() if p.requires_grad],
                          lr=0.0025, momentum=0.9, weight_decay=1e-4)

    def step(tm):
        t0 = time.perf_counter()
        opt.zero_grad()
        x = det.extract_feat(batch['img'])          # kd_one_stage.py:68
        t1 = time.perf_counter()
        with torch.no_grad():                       # :70-72
            teacher_x = det.teacher_model.extract_feat(batch['img'])
            out_teacher = det.teacher_model.bbox_head(teacher_x)
        t2 = time.perf_counter()
        outs = det.bbox_head(x)                     # ld_head.py:95-97
        t3 = time.perf_counter()
        losses = det.bbox_head.loss(*outs, batch['gt_bboxes'], batch['gt_labels'],
                                    out_teacher, x, teacher_x, batch['img_metas'])
        loss, _ = det._parse_losses(losses)         # base.py:185-218
        t4 = time.perf_counter()
        loss.backward()
        t5 = time.perf_counter()
        opt.step()
        t6 = time.perf_counter()
        tm.update(student_net=(t1 - t0) + (t3 - t2), teacher_net=t2 - t1,
                  loss_block=t4 - t3, backward=t5 - t4, optimizer=t6 - t5,
                  total=t6 - t0, loss=float(loss))

    step({})  # warm-up
    runs = []
    for _ in range(args.reps):
        tm = {}
        step(tm)
        runs.append(tm)
    runs.sort(key=lambda r: r['total'])
    med = runs[len(runs) // 2]
    print(json.dumps(dict(images=2, threads=args.threads, reps=args.reps,
                          stages_s={k: round(v, 4) for k, v in med.items()},
                          all_totals_s=[round(r['total'], 3) for r in runs])))


if __name__ == '__main__':
    main()
