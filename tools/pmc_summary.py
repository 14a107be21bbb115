"""Per-launch averages of rocprofv3 --pmc counter_collection.csv files for the
kernels whose name contains a substring:
    python tools/pmc_summary.py <substring> <csv> [<csv> ...]"""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
sub = sys.argv[1]
tot = collections.defaultdict(float)
cnt = collections.Counter()
dur = []
for path in sys.argv[2:]:
    seen = set()
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            if sub not in r['Kernel_Name']:
                continue
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            cnt[r['Counter_Name']] += 1
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in sorted(tot):
    print(f'{k:34s} {tot[k] / cnt[k]:16.1f}  (n={cnt[k]})')
if dur:
    print(f'duration_ns (under counters)       {sum(dur) / len(dur):16.1f}')
