"""Per-launch averages of rocprofv3 --pmc counter_collection.csv files for the
kernels whose name contains a substring ('a|b': either):
    python tools/pmc_summary.py <substring> <csv> [<csv> ...]
With PMC_BY_KERNEL=1 in the environment: additionally one line per kernel name
(launch count, per-launch average and total of every counter)."""
import collections
import csv
import os
import re
import sys

csv.field_size_limit(1 << 30)
sub = sys.argv[1]
tot = collections.defaultdict(float)
cnt = collections.Counter()
per = collections.defaultdict(lambda: collections.defaultdict(float))
pern = collections.defaultdict(collections.Counter)
dur = []
for path in sys.argv[2:]:
    seen = set()
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            if not any(a in r['Kernel_Name'] for a in sub.split('|')):
                continue
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            cnt[r['Counter_Name']] += 1
            short = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r['Kernel_Name'])
            per[short][r['Counter_Name']] += float(r['Counter_Value'])
            pern[short][r['Counter_Name']] += 1
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in sorted(tot):
    print(f'{k:34s} {tot[k] / cnt[k]:16.1f}  (n={cnt[k]})')
if dur:
    print(f'duration_ns (under counters)       {sum(dur) / len(dur):16.1f}')
if os.environ.get('PMC_BY_KERNEL') == '1':
    print('# per kernel: name, launches, then counter = per-launch average (total)')
    for name in sorted(per, key=lambda n: -sum(per[n].values())):
        parts = [f'{c} = {per[name][c] / pern[name][c]:.1f} ({per[name][c]:.0f})'
                 for c in sorted(per[name])]
        n = max(pern[name].values())
        print(f'{name[:90]:90s} n={n:5d}  ' + '  '.join(parts))
