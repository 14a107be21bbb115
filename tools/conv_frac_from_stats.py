"""Sum the conv kernels of a rocprofv3 --kernel-trace --stats CSV (the
*_kernel_stats.csv of a tools/profile_step.py --serial run) and compare with the
algorithmic conv FLOPs of the C2 step: reproduces bench.py's roofline.frac from
the committed profile alone.

    python tools/conv_frac_from_stats.py profiles/r04_rocprof_kernel_stats_fp32_serial.csv --steps 8
"""
import argparse
import csv

PEAK = 157.3
GFLOP_PER_STEP = 3653.5  # bench.py roofline.gflop_per_step, C2 (2 x 1826.75)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, required=True,
                    help='train steps inside the traced region (warm-up + timed)')
    args = ap.parse_args()
    conv = other = 0.0
    rows = []
    for r in csv.DictReader(open(args.csv)):
        ns = float(r['TotalDurationNs'])
        name = r['Name']
        is_conv = ('conv_' in name and 'weight_transform' not in name and
                   'to_c8' not in name)
        if is_conv:
            conv += ns
            rows.append((ns, int(r['Calls']), name))
        else:
            other += ns
    ms = conv / args.steps * 1e-6
    print(f'conv kernels: {ms:.3f} ms/step over {args.steps} steps '
          f'({sum(c for _, c, _ in rows) / args.steps:.0f} launches/step); '
          f'other kernels {other / args.steps * 1e-6:.3f} ms/step')
    print(f'{GFLOP_PER_STEP} GFLOP / {ms:.3f} ms = {GFLOP_PER_STEP / ms:.1f} TFLOP/s = '
          f'{GFLOP_PER_STEP / ms / PEAK:.3f} of {PEAK} TFLOP/s')
    for ns, calls, name in sorted(rows, reverse=True)[:12]:
        print(f'  {ns / args.steps * 1e-6:8.3f} ms/step  {calls / args.steps:6.1f}/step  {name[:110]}')


if __name__ == '__main__':
    main()
