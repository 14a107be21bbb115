"""Per-queue occupancy of the steady-state train step from a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv -- python tools/profile_step.py
--mode M --pipeline ...`): for the last N steps (delimited by the optimizer
launch), per hardware queue the number of dispatches, the summed kernel time and
the time the queue had a kernel running (union of intervals), the time NO queue
had a kernel running, and per queue the largest kernels.  Says which stream is
the critical path of the overlapped step and how much of the step the GPU idles.

    python tools/queue_busy.py step_kernel_trace.csv [--steps 5]
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:70]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--marker', default='sgd_kernel')
    ap.add_argument('--top', type=int, default=6)
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                         r['Queue_Id'], r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if args.marker in r[3]]
    marks = marks[-(args.steps + 1):]
    sel = rows[marks[0]:marks[-1]]
    span = rows[marks[-1]][0] - rows[marks[0]][0]
    n = args.steps
    print(f'# last {n} steps: {span / n / 1e6:.3f} ms per step under the trace')
    allbusy = union([(s, e) for s, e, _, _ in sel])
    print(f'GPU busy (any queue) {allbusy / n / 1e6:.3f} ms per step = '
          f'{100 * allbusy / span:.1f} %; idle {(span - allbusy) / n / 1e6:.3f} ms')
    byq = collections.defaultdict(list)
    for s, e, q, name in sel:
        byq[q].append((s, e, name))
    for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = union([(s, e) for s, e, _ in lst])
        tot = sum(e - s for s, e, _ in lst)
        print(f'queue {q}: {len(lst) / n:.1f} dispatches / step, kernel time '
              f'{tot / n / 1e6:.3f} ms, busy {busy / n / 1e6:.3f} ms = '
              f'{100 * busy / span:.1f} % of the step')
        # gaps between consecutive kernels of this queue
        lst.sort()
        gaps = [b[0] - a[1] for a, b in zip(lst[:-1], lst[1:]) if b[0] > a[1]]
        if gaps:
            gaps.sort()
            big = [g for g in gaps if g > 20000]
            print(f'    gaps: median {gaps[len(gaps) // 2] / 1e3:.1f} us, '
                  f'{len(big) / n:.1f} per step above 20 us totalling '
                  f'{sum(big) / n / 1e6:.3f} ms per step')
        dur = collections.Counter()
        cnt = collections.Counter()
        for s, e, name in lst:
            dur[short(name)] += e - s
            cnt[short(name)] += 1
        for name, d in dur.most_common(args.top):
            print(f'    {cnt[name] / n:6.1f} x  {d / n / 1e3:9.1f} us  {name}')


if __name__ == '__main__':
    main()
