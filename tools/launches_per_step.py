"""Launches of ONE steady-state train step, counted from a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv ... -- python tools/profile_step.py
...`): the dispatches between the starts of two consecutive optimizer launches
(`sgd_kernel`, exactly one per step), for the last few steps of the run.  Unlike
the `Calls` column of `--stats` divided by a step count this leaves out model
set-up, tuning and warm-up launches.

    python tools/launches_per_step.py step_kernel_trace.csv [--steps 5] > out.txt
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=5,
                    help='average over the last N complete steps')
    ap.add_argument('--marker', default='sgd_kernel')
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                         r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
    if len(marks) < args.steps + 1:
        raise SystemExit(f'only {len(marks)} {args.marker} launches in the trace')
    marks = marks[-(args.steps + 1):]
    counts = [b - a for a, b in zip(marks[:-1], marks[1:])]
    print(f'# dispatches between consecutive {args.marker} launches, last '
          f'{args.steps} steps: {counts}')
    print(f'launches_per_step {sum(counts) / len(counts):.1f}')
    span = rows[marks[-1]][0] - rows[marks[0]][0]
    print(f'ms_per_step_under_trace {span / args.steps / 1e6:.3f}')
    by = collections.Counter()
    dur = collections.Counter()
    for s, e, n in rows[marks[0]:marks[-1]]:
        by[short(n)] += 1
        dur[short(n)] += e - s
    conv = sum(c for n, c in by.items() if n.startswith('conv_'))
    print(f'of which conv_* kernels {conv / args.steps:.1f}, '
          f'runtime copy / fill kernels '
          f'{sum(c for n, c in by.items() if "rocclr" in n) / args.steps:.1f}, '
          f'ATen kernels '
          f'{sum(c for n, c in by.items() if n.startswith("at::")) / args.steps:.1f}')
    print('# per step: launches, us, kernel')
    for n, c in sorted(by.items(), key=lambda kv: -dur[kv[0]]):
        print(f'{c / args.steps:8.1f} {dur[n] / args.steps / 1e3:10.1f}  {n}')


if __name__ == '__main__':
    main()
