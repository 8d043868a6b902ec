"""bench.py's own multi-rank path on CPU: `python bench.py --gpus N --dry-run` must start N ranks itself (fresh child
processes, gloo rendezvous on 127.0.0.1), shard the global env ids with sharding.shard_range, reduce over all ranks and print
ONE line with n_gpus == N.  The dry run swaps the env for a stub (no GPU here) but runs the same Rank / shard_of /
timed_steps / reduce code the measured run uses.  A launcher that silently falls back to one rank fails these tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'TORCHELASTIC_RUN_ID', 'SGX_BENCH_LAUNCHER')}
    env.update(extra)
    return env


def _run(args, env=None, timeout=240):
    p = subprocess.run([sys.executable, BENCH] + args, env=env or _clean_env(), capture_output=True, text=True, timeout=timeout)
    return p


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith('{')]


def test_self_launch_two_ranks():
    p = _run(['--gpus', '2', '--dry-run', '--steps', '5', '--warmup', '2', '--envs', '1000'])
    assert p.returncode == 0, p.stderr
    lines = _json_lines(p.stdout)
    assert len(lines) == 1                                    # rank 0 only
    d = lines[0]
    assert d['n_gpus'] == 2 and d['dry_run'] is True and d['value'] is None
    assert d['launched_by'] == 'bench.py' and d['scaling'] == 'weak'
    assert d['config']['total_games'] == 2000 and d['config']['games_covered_by_ranks'] == 2000
    assert d['config']['stub_steps_x_games'] == 5 * 2000      # the SUM all-reduce saw both ranks' timed steps


def test_nccl_that_cannot_come_up_falls_back_to_gloo_and_says_so():
    """RCCL carries only the reporting reductions (the games never interact).  If the nccl process group cannot come up -- here: no GPU at
    all -- every rank falls back to gloo on a fresh store, the run completes, and the line names what happened."""
    p = _run(['--gpus', '2', '--dry-run', '--backend', 'nccl', '--steps', '4', '--warmup', '1', '--envs', '500'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d['n_gpus'] == 2 and d['config']['stub_steps_x_games'] == 4 * 1000
    assert d['config']['reduction_backend'].startswith('gloo (nccl failed: '), d['config']['reduction_backend']
    assert 'fall back to gloo' in p.stderr


def test_self_launch_eight_ranks_with_the_config5_defaults():
    """BASELINE config 5 as the driver will launch it: --gpus 8 with no size argument = 262,144 games per GPU (2,097,152 in total,
    weak) plus the strong leg of 2,097,152 games; all eight ranks rendezvous, shard and reduce."""
    p = _run(['--gpus', '8', '--dry-run', '--steps', '3', '--warmup', '1'], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['launched_by'] == 'bench.py'
    assert d['config']['games_per_gpu'] == 262144 and d['config']['total_games'] == 2097152
    assert d['config']['games_covered_by_ranks'] == 2097152 and d['config']['stub_steps_x_games'] == 3 * 2097152
    assert d['config']['strong_leg_total_games'] == 2097152
    # the strong split of the same total over 8 ranks, and one that does not divide evenly
    p = _run(['--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '1', '--total-envs', '2097152'], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d['scaling'] == 'strong' and d['config']['games_per_gpu'] == 262144 and d['config']['games_covered_by_ranks'] == 2097152
    p = _run(['--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '1', '--total-envs', '65541'], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d['config']['games_covered_by_ranks'] == 65541 and d['config']['stub_steps_x_games'] == 2 * 65541


def test_self_launch_strong_scaling_keeps_the_remainder():
    # 1001 games over 3 ranks: shard_range gives 334 + 334 + 333; an integer division would drop two games
    p = _run(['--gpus', '3', '--dry-run', '--steps', '4', '--warmup', '1', '--total-envs', '1001'])
    assert p.returncode == 0, p.stderr
    d = _json_lines(p.stdout)[0]
    assert d['n_gpus'] == 3 and d['scaling'] == 'strong'
    assert d['config']['games_covered_by_ranks'] == 1001 and d['config']['stub_steps_x_games'] == 4 * 1001


def test_external_launcher_world_size_must_match_gpus():
    # a launcher that started ONE rank for --gpus 2 is an error, not a 1-GPU run
    p = _run(['--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'], env=_clean_env(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0'))
    assert p.returncode != 0 and 'WORLD_SIZE=1' in (p.stderr + p.stdout)
    # and the other way round: --gpus 1 inside a 2-rank job
    p = _run(['--gpus', '1', '--dry-run', '--steps', '2', '--warmup', '1'], env=_clean_env(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0'))
    assert p.returncode != 0 and 'WORLD_SIZE=2' in (p.stderr + p.stdout)


def test_under_torch_distributed_run():
    """The driver's launch form: python -m torch.distributed.run ... bench.py --gpus N (ranks from the environment)."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), BENCH, '--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1',
                        '--envs', '64'], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['launched_by'] == 'external'
    assert lines[0]['config']['games_covered_by_ranks'] == 128


def test_single_rank_dry_run_and_failing_rank_stops_the_job():
    p = _run(['--gpus', '1', '--dry-run', '--steps', '3', '--warmup', '1', '--envs', '10'])
    assert p.returncode == 0, p.stderr
    assert _json_lines(p.stdout)[0]['n_gpus'] == 1
    # rank 1 of 2 gets no games (1 game in total): every rank must exit non-zero instead of hanging in the barrier
    p = _run(['--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1', '--total-envs', '1'], timeout=120)
    assert p.returncode != 0 and not _json_lines(p.stdout)


def test_measured_run_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = _run(['--gpus', '1', '--steps', '2', '--warmup', '1'])
    assert p.returncode != 0 and 'no CPU fallback' in (p.stderr + p.stdout)
    p = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'])          # the launcher counts devices before starting ranks
    assert p.returncode != 0 and 'GPU(s) are visible' in (p.stderr + p.stdout)


def test_defaults_follow_the_baseline_configs_and_the_parent_never_loads_torch():
    """--gpus 1: 65,536 games (BASELINE config 2); --gpus N > 1: 262,144 games per GPU (config 5: 2,097,152 on 8 GPUs) plus the
    strong-scaling leg of 2,097,152 games in total; the launcher parent counts GPUs from sysfs without importing torch."""
    p = _run(['--gpus', '1', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert p.returncode == 0, p.stderr
    d = _json_lines(p.stdout)[0]
    assert d['config']['games_per_gpu'] == 65536 and d['config']['strong_leg_total_games'] == 0
    p = _run(['--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert p.returncode == 0, p.stderr
    d = _json_lines(p.stdout)[0]
    assert d['config']['games_per_gpu'] == 262144 and d['config']['total_games'] == 524288
    assert d['config']['strong_leg_total_games'] == 2097152
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpus(); "
            "assert isinstance(n, int) and n >= 0; assert 'torch' not in sys.modules; print('ok', n)" % ROOT)
    q = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert q.returncode == 0 and q.stdout.startswith('ok'), q.stderr
