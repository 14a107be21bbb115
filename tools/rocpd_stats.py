"""Per-kernel statistics (calls, total / average / min / max duration) from a
rocprofv3 rocpd SQLite database -- ROCm 7's default output of
`rocprofv3 --kernel-trace --stats` -- as the CSV this repo keeps under
profiles/.  usage: python tools/rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        'select s.kernel_name, count(*), sum(d.end - d.start), '
        'avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) '
        'from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
        'on d.kernel_id = s.id group by s.kernel_name order by 3 desc'))
    total = sum(r[2] for r in rows) or 1
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs',
                    'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1),
                        round(100 * r[2] / total, 3), r[4], r[5]])
    print(f'{len(rows)} kernels, {total / 1e6:.2f} ms of kernel time -> {out}')


if __name__ == '__main__':
    main(*sys.argv[1:3])
