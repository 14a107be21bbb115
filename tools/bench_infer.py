"""Latency of the device-side GFLHead.get_bboxes (ld_get_bboxes) at the C2 size
(2 x 800x1344, 22400 anchors/img, 80 classes).  Writes
gpurun_out/infer_<tag>.json."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ld_amd import lossblock as LB, synthetic  # noqa: E402

dev = torch.device('cuda:0')
out = dict(device=torch.cuda.get_device_name(0), time=time.time(), cases=[])
for case in synthetic.INFER_CASES:
    if not case[0].startswith('c2'):
        continue
    cls, reg, metas = synthetic.infer_inputs(case, device=dev)
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas]

    def run():
        return LB.get_bboxes(cls, reg, (8, 16, 32, 64, 128), shapes, sfs,
                             nms_pre=case[5])

    for _ in range(3):
        res = run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        res = run()  # ends with the D2H read of the counts: synchronous
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    N = cls[0].shape[0]
    A = sum(c.shape[2] * c.shape[3] for c in cls)
    bytes_in = N * A * (80 + 68) * 4
    out['cases'].append(dict(
        case=case[0], ms_per_batch=med * 1e3, images=N,
        images_per_s=N / med, dets=[int(d.shape[0]) for d, _ in res],
        algorithmic_read_MB=bytes_in / 1e6,
        note='wall time of the whole call incl. the count read-back'))
    print(out['cases'][-1], flush=True)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
with open(os.path.join(REPO, 'gpurun_out', f'infer_{tag}.json'), 'w') as f:
    json.dump(out, f, indent=1)
