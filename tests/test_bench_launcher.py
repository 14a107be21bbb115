"""bench.py --gpus N must run on N ranks or refuse: the
launcher logic, on CPU, with a stubbed device count."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_launch_plan_decisions():
    import bench
    assert bench.launch_plan(1, {}, 1, []) == ('run', None)
    # a launcher with the right rank count: run
    assert bench.launch_plan(8, {'WORLD_SIZE': '8', 'LOCAL_RANK': '7'}, 8,
                             []) == ('run', None)
    # no launcher: re-execute as N ranks, arguments passed through
    action, cmd = bench.launch_plan(8, {}, 8, ['--gpus', '8', '--steps', '5'])
    assert action == 'spawn'
    assert '--nproc-per-node=8' in cmd and cmd[-4:] == ['--gpus', '8', '--steps', '5']
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    # refusals: a line that says n_gpus: N must have run on N GPUs
    with pytest.raises(SystemExit, match='only 1 GPU'):
        bench.launch_plan(8, {}, 1, [])
    with pytest.raises(SystemExit, match='only 1 GPU'):
        bench.launch_plan(8, {'WORLD_SIZE': '8'}, 1, [])
    with pytest.raises(SystemExit, match='WORLD_SIZE=2'):
        bench.launch_plan(8, {'WORLD_SIZE': '2'}, 8, [])
    with pytest.raises(SystemExit, match='WORLD_SIZE=2'):
        bench.launch_plan(1, {'WORLD_SIZE': '2'}, 8, [])
    with pytest.raises(SystemExit, match='no device'):
        bench.launch_plan(2, {'WORLD_SIZE': '2', 'LOCAL_RANK': '3'}, 2, [])


def test_bench_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment starts two
    ranks (torch.distributed.run on 127.0.0.1); with one visible device it
    exits non-zero and says why."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(LD_BENCH_FAKE_DEVICES='2', LD_BENCH_LAUNCH_ONLY='1')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2',
                        '--steps', '1', '--warmup', '0'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout + r.stderr
    assert 'launch-only rank 0 of 2' in out and 'launch-only rank 1 of 2' in out
    env['LD_BENCH_FAKE_DEVICES'] = '1'
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2'],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert 'only 1 GPU' in r.stderr
