"""CPU-side guard for two compiler effects found in round 3 by reading the ISA
(DESIGN.md section 3.3 / 3.4 "late"): the pipelined loops of the bf16 kernels must
wait with COUNTED vmcnt values.  A `vmcnt(0)` inside the ring loop means the
loads are not running ahead (a branch around them makes hipcc merge the memory
counters of both paths).  hipcc cross-compiles for gfx950 without a GPU."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))


@pytest.fixture(scope='module')
def bf16_rows():
    import isa_lint
    src = os.path.join(REPO, 'ld_amd', 'csrc', 'conv_bf16.hip')
    if not os.path.exists(isa_lint.os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    return isa_lint.lint(isa_lint.device_asm(src))


def _loops(rows, needle):
    hit = [(n, h) for n, h, _ in rows if needle in n and h]
    assert hit, f'no kernel matching {needle!r} has a loop with vmcnt waits'
    return hit


def test_c8_weight_gradient_rings_prefetch(bf16_rows):
    """Both C8 weight-gradient kernels: the LDS writes of step u wait for the
    ring slot loaded two steps earlier, not for the loads just issued."""
    for needle, floor in (('conv_wgrad_c8_kernelILi3', 16),
                          ('conv_wgrad_c8_tile_kernelILi4', 8)):
        for name, hist in _loops(bf16_rows, needle):
            deep = sum(c for v, c in hist.items() if v >= floor)
            assert deep >= 8, (name, hist)
            assert hist.get(0, 0) <= 1, (name, hist)  # at most the drain wait


def test_c8_tiled_forward_rings_prefetch(bf16_rows):
    """Every shape of the C8 tiled forward / data-gradient kernel waits with a
    non-zero count inside its k-loop."""
    seen = 0
    for name, hist in _loops(bf16_rows, 'conv_tile_c8_kernel'):
        assert min(hist) >= 2, (name, hist)
        seen += 1
    assert seen >= 40  # 22 shapes x 2 modes


@pytest.fixture(scope='module')
def fp32_rows():
    import isa_lint
    src = os.path.join(REPO, 'ld_amd', 'csrc', 'conv.hip')
    if not os.path.exists(isa_lint.os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    return isa_lint.lint(isa_lint.device_asm(src))


def test_fp32_stream_kernel_loop_and_epilogue(fp32_rows):
    """The fp32 streaming conv: counted waits in the k-loop, and an epilogue
    that is not a chain of dependent loads.  The r-major epilogue over may-alias
    pointers this round replaced had ~155 `vmcnt(0)` per kernel (one per affine
    / residual load); the grouped, descriptor-based one has a few per row group
    and variant."""
    seen = 0
    for name, hist, outside0 in fp32_rows:
        if 'conv_stream_kernel' not in name or not hist:
            continue
        seen += 1
        assert 0 not in hist or hist[0] <= 1, (name, hist)
        assert outside0 <= 48, (name, outside0)
    assert seen >= 50


@pytest.fixture(scope='module')
def wgrad_asm():
    import isa_lint
    src = os.path.join(REPO, 'ld_amd', 'csrc', 'conv_wgrad.hip')
    if not os.path.exists(isa_lint.os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    return isa_lint.device_asm(src)


def test_fp32_tiled_weight_gradient_loops_and_registers(wgrad_asm):
    """Round 4, conv_wgrad.hip: the staged slice written to LDS in step u was
    loaded a whole slice earlier -- the waits inside the main loops of the tiled
    and the three-taps kernels are COUNTED (never vmcnt(0)); no shipped instance
    spills (the in-launch combination's tail did before it was restructured), and
    the (kg 2, bk 32) instance stays at <= 128 VGPRs = four waves per SIMD."""
    import isa_lint
    import re
    rows = isa_lint.lint(wgrad_asm)
    seen = 0
    for name, hist, _ in rows:
        m = re.search(r'conv_wgrad_(tile|tap3)_kernelI(\S*?)v6WgradK', name)
        if not m or not hist:
            continue
        targs = [int(v) for v in re.findall(r'L[ib](\d+)E', m.group(2))]
        if targs[-1] != 0:
            continue  # LD_WGRAD_DBG timing-attribution variants
        seen += 1
        # (the lint pools every loop of a kernel: the k-group / in-launch
        # combination tails legitimately drain with vmcnt(0); the MAIN loop shows
        # as a population of deep counted waits)
        deep = sum(c for v, c in hist.items() if v >= 6)
        assert deep >= 6 and max(hist) >= 8, (name, hist)  # (kg 4, bk 32): 8 loads per slice
        if m.group(1) == 'tap3':
            assert 0 not in hist, (name, hist)  # no combination tail there
    assert seen >= 12, seen  # 5 (kg, bk) x 2 level modes + 2 tap3
    # resource audit from the code object metadata
    meta = re.findall(r'\.name:\s+(\S*conv_wgrad_(?:tile|tap3)_kernel\S*)\n(?:.*\n)*?'
                      r'\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?'
                      r'\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', wgrad_asm)
    assert len(meta) >= 12
    for name, scratch, vgpr, spill in meta:
        targs = [int(v) for v in re.findall(r'L[ib](\d+)E', name.split('kernelI')[1])]
        if targs[-1] != 0:
            continue  # timing-attribution variants
        assert int(spill) == 0 and int(scratch) == 0, (name, scratch, spill)
        if 'ILi2ELi32E' in name:
            assert int(vgpr) <= 128, (name, vgpr)
        if 'tap3' in name:
            assert int(vgpr) <= 168, (name, vgpr)  # three waves per SIMD


@pytest.fixture(scope='module')
def fused_asm():
    import isa_lint
    src = os.path.join(REPO, 'ld_amd', 'csrc', 'conv_fused.hip')
    if not os.path.exists(isa_lint.os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    return isa_lint.device_asm(src)


def test_fused_bottleneck_rings_prefetch_and_no_spill(fused_asm):
    """Round 5, conv_fused.hip: the first version of the fused teacher bottleneck
    lost its prefetch distance twice while it was written -- hipcc sank every
    weight load below the MFMAs (one vmcnt(0) per step) until the ring was pinned
    with scheduling barriers, and a 4-slot ring in G2 / G3 covered only 512 clk.
    The shipped kernel waits with COUNTED vmcnt values (ring depth 8 in G2 / G3:
    waits of 13-15 outstanding loads dominate), drains with vmcnt(0) only at the
    pass ends, keeps its accumulators in AGPRs and does not spill."""
    import isa_lint
    import re
    rows = [(n, h) for n, h, _ in isa_lint.lint(fused_asm)
            if 'fused_bottleneck_c8_kernel' in n and h]
    assert len(rows) == 1, [n for n, _ in rows]
    name, hist = rows[0]
    deep = sum(c for v, c in hist.items() if v >= 12)
    assert deep >= 40, (name, hist)
    assert hist.get(0, 0) <= 6, (name, hist)
    meta = re.findall(r'\.name:\s+(\S*fused_bottleneck_c8_kernel\S*)\n(?:.*\n)*?'
                      r'\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?'
                      r'\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)',
                      fused_asm)
    assert len(meta) == 1
    _, scratch, vgpr, spill = meta[0]
    assert int(spill) == 0 and int(scratch) == 0, meta
    assert int(vgpr) <= 512, meta  # unified VGPR + AGPR file of one wave per SIMD
    agpr = re.findall(r'\.agpr_count:\s+(\d+)', fused_asm)
    assert agpr and max(int(a) for a in agpr) >= 128, agpr


@pytest.fixture(scope='module')
def t256_asm():
    import isa_lint
    src = os.path.join(REPO, 'ld_amd', 'csrc', 'conv_t256.hip')
    if not os.path.exists(isa_lint.os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    return isa_lint.device_asm(src)


def test_t256_lds_dma_loop_shape_and_registers(t256_asm):
    """Round 6, conv_t256.hip: every instance of the 8-wave LDS-DMA kernel keeps the
    loop it was designed with -- operands arrive by `buffer_load_dwordx4 ... lds`
    (never through VGPRs + ds_write), the ONLY vmcnt wait of the main loop is the
    one in front of the step's barrier (hipcc must not add one in front of the
    fragment reads: a second __shared__ object or a VGPR-destination load in the
    loop makes it), the barrier count equals the unrolled step count, and nothing
    spills with 128 / 96 accumulator registers."""
    import re
    names = re.findall(r'^(_Z\w*conv_t256_c8_kernel\w*):\s', t256_asm, re.M)
    assert len(names) >= 12, names  # 4 tile shapes x (plain, swapped, stride-2 class)
    for name in names:
        body = t256_asm.split('\n' + name + ':', 1)[1].split('s_endpgm', 1)[0]
        lines = body.split('\n')
        mf = [i for i, l in enumerate(lines) if 'v_mfma_f32_32x32x16_bf16' in l]
        loop = lines[mf[0]:mf[-1] + 1]
        dma = [l for l in loop if 'buffer_load_dwordx4' in l and ' lds' in l]
        assert len(dma) >= 8, (name, len(dma))  # two unrolled steps of >= 4 per wave
        assert not any('ds_write' in l for l in loop), name
        assert not any(l.strip().startswith(('global_load', 'flat_load')) for l in loop), name
        waits = [l for l in loop if 's_waitcnt' in l and 'vmcnt' in l]
        bars = [l for l in loop if l.strip().startswith('s_barrier')]
        assert len(bars) == 2 and len(waits) == 2, (name, waits, bars)
        assert all('vmcnt(0)' in w and 'lgkmcnt(0)' in w for w in waits), (name, waits)
    meta = re.findall(r'\.name:\s+(\S*conv_t256_c8_kernel\S*)\n(?:.*\n)*?'
                      r'\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?'
                      r'\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t256_asm)
    assert len(meta) >= 12
    for name, scratch, vgpr, spill in meta:
        assert int(spill) == 0 and int(scratch) == 0, (name, scratch, spill)
        assert int(vgpr) <= 256, (name, vgpr)  # two waves per SIMD
