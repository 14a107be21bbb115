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
