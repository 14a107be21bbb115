"""Host logic added in round 5 that needs no GPU: the runtime-configuration check
for hipGraph replays, the refusal of captures that would contain collectives, the
stepper's default, the launch-count / queue-occupancy tools on a synthetic kernel
trace, the odd-repetition rule of the CPU baseline, and bench.py's single
traffic ratio."""
import csv
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('hwq,gq,ok', [
    (None, None, True),    # runtime defaults: 4 hardware queues
    ('4', None, True),
    ('8', None, False),    # 4 internal graph streams on 8 queues: the slow replay
    ('8', '2', True),      # what `import ld_amd` sets in a multi-process job
    ('8', '4', False),
    ('2', '4', True),
    ('x', None, True),     # unparsable -> the runtime's default
])
def test_graph_queues_ok(monkeypatch, hwq, gq, ok):
    from ld_amd import train
    for name, v in (('GPU_MAX_HW_QUEUES', hwq),
                    ('DEBUG_HIP_FORCE_GRAPH_QUEUES', gq)):
        if v is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, v)
    assert train.graph_queues_ok() is ok


def test_capture_with_collectives_is_refused(tmp_path, monkeypatch):
    """A one-rank gloo group with the collectives forced stands in for the
    multi-process job: a step capture must raise, LD_GRAPH_COLLECTIVES=1 is the
    explicit opt-in, and without a group nothing is refused."""
    from ld_amd import train
    monkeypatch.delenv('LD_GRAPH_COLLECTIVES', raising=False)
    train._refuse_collectives_in_capture('no group')  # no process group: fine
    assert train._capture_mode() == 'global'
    dist.init_process_group('gloo', init_method=f'file://{tmp_path}/pg',
                            rank=0, world_size=1)
    try:
        monkeypatch.setenv('LD_FORCE_COLLECTIVES', '1')
        assert train.collectives_on()
        assert train._capture_mode() == 'thread_local'
        with pytest.raises(RuntimeError, match='collectives is refused'):
            train._refuse_collectives_in_capture('GraphedStep')
        with train.suspend_collectives():  # a capture's warm-up window
            train._refuse_collectives_in_capture('warm-up')
        monkeypatch.setenv('LD_GRAPH_COLLECTIVES', '1')
        train._refuse_collectives_in_capture('opt-in')
        monkeypatch.setenv('LD_FORCE_COLLECTIVES', '0')
        monkeypatch.delenv('LD_GRAPH_COLLECTIVES')
        assert not train.collectives_on()  # one rank, nothing forced
        train._refuse_collectives_in_capture('one rank')
    finally:
        dist.destroy_process_group()


def test_auto_stepper_defaults_to_eager():
    from ld_amd.train import AutoStepper

    class _Trainer:
        model = torch.nn.Linear(2, 2)

    assert AutoStepper(_Trainer()).mode == 'eager'
    assert AutoStepper(_Trainer(), mode='graph').mode == 'graph'
    with pytest.raises(ValueError):
        AutoStepper(_Trainer(), mode='fast')


def test_auto_stepper_launcher_and_bench_rank_pinning(monkeypatch):
    """Round 6 host logic: the stepper's launcher option, and bench.pin_rank giving
    the ranks of one node disjoint, contiguous core slices + a small OpenMP pool."""
    from ld_amd.train import AutoStepper

    class _Trainer:
        model = torch.nn.Linear(2, 2)

    assert AutoStepper(_Trainer(), mode='graph', launcher='list').launcher == 'list'
    with pytest.raises(ValueError):
        AutoStepper(_Trainer(), mode='graph', launcher='fast')
    sys.path.insert(0, REPO)
    import bench
    cores = set(range(64))
    pinned = {}
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(cores))
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, s: pinned.update(last=set(s)))
    monkeypatch.setattr(torch, 'set_num_threads', lambda n: None)
    monkeypatch.delenv('OMP_NUM_THREADS', raising=False)
    seen = []
    for local in range(8):
        monkeypatch.delenv('OMP_NUM_THREADS', raising=False)
        r = bench.pin_rank(local, 8)
        assert r['cpus'] == 8 and r['omp_threads'] == 8
        assert r['first'] == local * 8 and r['last'] == local * 8 + 7
        seen.append(pinned['last'])
    assert set().union(*seen) == cores and sum(len(s) for s in seen) == 64
    monkeypatch.setenv('LD_BENCH_PIN', '0')
    assert bench.pin_rank(0, 8) is None


def test_bench_pmc_constants_match_the_committed_profiles():
    """bench.PMC_CONV (roofline.traffic / roofline_bf16.traffic) == what
    tools/pmc_conv_bytes.py derives from the committed by-kernel PMC summaries."""
    import re
    sys.path.insert(0, REPO)
    import bench
    for mode, steps in (('fp32', 6), ('bf16', 7)):
        c = bench.PMC_CONV[mode]
        out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pmc_conv_bytes.py'),
                              os.path.join(REPO, c['file']), str(steps)],
                             capture_output=True, text=True, check=True).stdout
        disp = float(re.search(r'= ([\d.]+) per step', out).group(1))
        fetch = float(re.search(r'FETCH_SIZE x 2 = ([\d.]+) GB', out).group(1))
        write = float(re.search(r'WRITE_SIZE     = ([\d.]+) GB', out).group(1))
        assert disp == c['dispatches']
        assert abs(fetch * 1e9 - c['fetch']) < 2e6 and abs(write * 1e9 - c['write']) < 2e6


def _write_trace(path, steps=4):
    """Two queues; per step: 3 conv + 1 norm kernel on queue 1, 2 conv on queue 2,
    then the optimizer launch.  10 us kernels, 5 us gaps on queue 1."""
    cols = ['Kind', 'Agent_Id', 'Queue_Id', 'Stream_Id', 'Thread_Id',
            'Dispatch_Id', 'Kernel_Id', 'Kernel_Name', 'Correlation_Id',
            'Start_Timestamp', 'End_Timestamp']
    rows, t, n = [], 1_000_000, 0
    for _ in range(steps + 2):
        names = [('1', 'void (anonymous namespace)::conv_stream_kernel<1, 1>(ConvK)'),
                 ('2', 'void (anonymous namespace)::conv_wgrad_tile_kernel<2>(WgradK)'),
                 ('1', 'void (anonymous namespace)::conv_stream_kernel<1, 1>(ConvK)'),
                 ('1', '(anonymous namespace)::gn_apply_kernel(float const*)'),
                 ('2', 'void (anonymous namespace)::conv_wgrad_tile_kernel<2>(WgradK)'),
                 ('1', 'void (anonymous namespace)::conv_stream_kernel<3, 1>(ConvK)'),
                 ('1', '(anonymous namespace)::sgd_kernel(float*, float const*)')]
        for q, name in names:
            n += 1
            rows.append(dict(zip(cols, ['KERNEL_DISPATCH', 'Agent 2', q, 0, 1, n,
                                        1, name, n, t, t + 10_000])))
            t += 15_000 if q == '1' else 2_000
    with open(path, 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=cols, quoting=csv.QUOTE_ALL)
        w.writeheader()
        w.writerows(rows)


def test_launches_per_step_and_queue_busy_tools(tmp_path):
    trace = tmp_path / 'step_kernel_trace.csv'
    _write_trace(trace)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools',
                                                       'launches_per_step.py'),
                          str(trace), '--steps', '3'], capture_output=True,
                         text=True, check=True).stdout
    assert 'launches_per_step 7.0' in out
    assert 'of which conv_* kernels 5.0' in out
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools',
                                                       'queue_busy.py'),
                          str(trace), '--steps', '3'], capture_output=True,
                         text=True, check=True).stdout
    assert 'queue 1: 5.0 dispatches / step' in out
    assert 'queue 2: 2.0 dispatches / step' in out
    # 7 kernels of 10 us per step, the two of queue 2 overlap queue 1's
    assert 'kernel time 0.050 ms' in out and 'kernel time 0.020 ms' in out


def test_cpu_baseline_refuses_even_repetitions():
    """the CPU baseline is the MEDIAN of an odd number of
    timed steps."""
    r = subprocess.run([sys.executable, os.path.join(REPO, 'oracle',
                                                     'ref_cpu_step.py'),
                        '--reps', '2'], capture_output=True, text=True)
    assert r.returncode != 0
    assert 'odd' in (r.stderr + r.stdout)


def test_bench_traffic_fields_state_one_ratio(monkeypatch):
    """roofline.traffic / roofline.algorithmic_bytes_per_launch == the ratio the
    note quotes == traffic_over_algorithmic."""
    sys.path.insert(0, REPO)
    import bench
    from ld_amd import layers as Y

    class _Prof:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            pass

        def summary(self):
            return {'conv_fwd': (0.019, 2.0e12, 193), 'conv_dgrad': (0.007, 8e11, 58),
                    'conv_wgrad': (0.008, 8e11, 65)}

        def algorithmic_bytes(self):
            return 18.37e9

        def algorithmic_read_write(self):
            return (10.98e9, 7.39e9)

        def fused_read_write(self):
            return (3.19e9, 1.01e9)

    class _Trainer:
        class model:
            use_teacher_stream = False

        def step(self, d):
            pass

    monkeypatch.setattr(Y, 'KernelProfile', _Prof)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a: None)
    r = bench.kernel_roofline(_Trainer(), None, 1)
    ratio = r['traffic'] / r['algorithmic_bytes_per_launch']
    assert abs(ratio - r['traffic_over_algorithmic']) < 1e-9
    assert ('= %.2f x' % ratio) in r['traffic_note']
    assert 1.0 < r['write_over_writes_incl_fused'] < r['write_over_algorithmic_writes']
    assert 1.0 < r['fetch_over_reads_incl_fused'] < r['fetch_over_algorithmic_reads']
    # round 6: the bf16 leg carries its own PMC figures
    rb = bench.kernel_roofline(_Trainer(), None, 1, bf16=True)
    assert rb['traffic'] is None or rb['traffic'] > 0
    assert bench.PMC_CONV['bf16']['file'].startswith('profiles/r06_')
