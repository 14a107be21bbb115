"""Per-step HBM-side bytes of the conv GEMM kernels from a by-kernel PMC summary
(tools/pmc_traffic.sh with PMC_BY_KERNEL=1): FETCH_SIZE (x 2, the gfx950
correction calibrated in profiles/r04_pmc_calib_copy_*) and WRITE_SIZE, both in
KiB per launch in the file, summed over the GEMM kernels (forward / data-gradient
/ weight-gradient / fused bottleneck; NOT the slab reduce, the weight transforms)
and divided by the steps in the run -- the constants bench.py's `roofline.traffic`
is built from.
    python tools/pmc_conv_bytes.py profiles/rNN_pmc_traffic_conv_step_*_by_kernel.txt <steps>"""
import re
import sys

GEMM = ('conv_stream_kernel', 'conv_wgrad_tile_kernel', 'conv_wgrad_tap3_kernel',
        'conv_stem_kernel', 'conv_stem_lds_kernel', 'conv_igemm_kernel', 'conv_tile_c8_kernel',
        'conv_t256_c8_kernel', 'conv_wgrad_c8_tile_kernel', 'conv_wgrad_c8_kernel',
        'fused_bottleneck_c8_kernel', 'conv_stream_bf16_kernel', 'conv_tile_bf16_kernel',
        'conv_wgrad_wave_bf16_kernel', 'conv_wgrad_tile_bf16_kernel', 'conv_wgrad_kernel')


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    fetch = write = 0.0
    n = 0
    for ln in open(path):
        m = re.match(r'\s*(\S.*?)\s+n=\s*(\d+)\s+FETCH_SIZE = [\d.]+ \((\d+)\)\s+'
                     r'WRITE_SIZE = [\d.]+ \((\d+)\)', ln)
        if not m:
            continue
        name = m.group(1)
        if not any(k in name for k in GEMM):
            continue
        n += int(m.group(2))
        fetch += float(m.group(3))
        write += float(m.group(4))
    kib = 1024.0
    print(f'{path}: {n} GEMM dispatches in {steps} steps = {n / steps:.1f} per step')
    print(f'  FETCH_SIZE x 2 = {2 * fetch * kib / steps / 1e9:.3f} GB per step')
    print(f'  WRITE_SIZE     = {write * kib / steps / 1e9:.3f} GB per step')
    print(f'  total          = {(2 * fetch + write) * kib / steps / 1e9:.3f} GB per step')


if __name__ == '__main__':
    main()
