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
    """RCCL carries only the reporting reductions (the games never interact).  If the nccl group cannot come up -- here: no GPU at
    all -- every rank uses the host-side gloo group for them, the run completes, and the line names what happened."""
    p = _run(['--gpus', '2', '--dry-run', '--backend', 'nccl', '--steps', '4', '--warmup', '1', '--envs', '500'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d['n_gpus'] == 2 and d['config']['stub_steps_x_games'] == 4 * 1000
    assert d['config']['reduction_backend'].startswith('gloo (nccl failed: rank 0: '), d['config']['reduction_backend']
    assert 'ALL ranks use gloo' in p.stderr


def test_one_rank_failing_to_bring_up_nccl_moves_all_ranks_to_gloo():
    """The ranks AGREE on the backend of the reporting reductions before using it (class Rank: verdicts all-gathered over the host-side
    gloo group, which is always there).  A second gloo group stands in for RCCL on this CPU box (SGX_BENCH_FAKE_NCCL): with every rank
    fine the job reduces over it; with ONE rank failing -- before the collective bring-up, or inside it while the others wait for it
    until their timeout -- all of them end up on gloo, nobody hangs, and the line names the rank."""
    common = ['--gpus', '4', '--dry-run', '--backend', 'nccl', '--steps', '3', '--warmup', '1', '--envs', '100']
    p = _run(common, env=_clean_env(SGX_BENCH_FAKE_NCCL='1'))
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d['config']['reduction_backend'].startswith('nccl (simulated') and d['config']['stub_steps_x_games'] == 3 * 400
    p = _run(common, env=_clean_env(SGX_BENCH_FAKE_NCCL='1', SGX_BENCH_FAIL_NCCL_RANKS='2'))
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d['config']['reduction_backend'].startswith('gloo (nccl failed: rank 2: '), d['config']['reduction_backend']
    assert d['config']['games_covered_by_ranks'] == 400 and d['config']['stub_steps_x_games'] == 3 * 400
    assert p.stderr.count('ALL ranks use gloo') == 4                       # every rank took the same decision
    p = _run(common, env=_clean_env(SGX_BENCH_FAKE_NCCL='1', SGX_BENCH_FAIL_NCCL_STAGE2_RANKS='1', SGX_BENCH_NCCL_TIMEOUT='4'))
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d['config']['reduction_backend'].startswith('gloo (nccl failed: ') and 'rank 1: RuntimeError: simulated failure inside' in d['config']['reduction_backend']
    assert d['config']['stub_steps_x_games'] == 3 * 400 and p.stderr.count('ALL ranks use gloo') == 4


@pytest.mark.slow
def test_self_launch_eight_ranks_with_the_default_workloads_is_self_anchoring():
    """The multi-GPU line as the driver will launch it: --gpus 8 with no size argument = the 1-GPU line's workload on every GPU (65,536
    games per GPU, a ring of three output sets: `value` over 1 / 2 / 4 / 8 GPUs is one workload's weak-scaling curve), and BASELINE
    config 5 -- 262,144 games per GPU, 2,097,152 in total -- as scaling_legs[0]; the headline and every leg carry a solo anchor (rank 0
    alone, the others parked) and scaling_x = value / solo value.  One rank's nccl failure is simulated on top: all eight agree on gloo."""
    import time
    t0 = time.perf_counter()
    p = _run(['--gpus', '8', '--dry-run', '--backend', 'nccl', '--steps', '3', '--warmup', '1'],
             env=_clean_env(SGX_BENCH_FAKE_NCCL='1', SGX_BENCH_FAIL_NCCL_RANKS='5'), timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    # the whole 8-rank line -- start of the ranks, rendezvous, agreement on the backend after one rank's failure, headline, anchors, config 5
    # leg -- in well under two minutes of host time (on 8 CPUs shared by 8 ranks here): nothing in the launcher path waits for a timeout
    assert time.perf_counter() - t0 < 120
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    d = lines[0]
    c = d['config']
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['launched_by'] == 'bench.py'
    assert c['reduction_backend'].startswith('gloo (nccl failed: rank 5: ') and 0 <= c['reduction_bringup_s'] < 30
    assert c['legs_of_this_line'] == ['scaling_legs'] and d.get('cpu_baseline') is None
    assert c['games_per_gpu'] == 65536 and c['total_games'] == 8 * 65536 and c['output_sets'] == 3
    assert c['games_covered_by_ranks'] == 8 * 65536 and c['stub_steps_x_games'] == 3 * 8 * 65536
    assert c['solo']['games'] == 65536 and c['solo']['value'] > 0 and c['scaling_x'] > 0 and c['scaling_x_ideal'] == 8
    assert 0 < c['per_gpu_value_min_over_solo'] and c['strong_leg_total_games'] == 2097152
    legs = c['scaling_legs']
    assert len(legs) == 1                                                  # on 8 GPUs the strong split of 2,097,152 IS config 5's 262,144 per GPU
    leg = legs[0]
    assert leg['games_per_gpu'] == 262144 and leg['total_games'] == 2097152 and leg['scaling'] == 'weak' and leg['output_sets'] == 1
    assert leg['solo']['games'] == 262144 and leg['scaling_x'] > 0 and leg['per_gpu_value_min_over_solo'] > 0
    # the strong split of the same total over 8 ranks, and one that does not divide evenly
    p = _run(['--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '1', '--total-envs', '2097152'], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d['scaling'] == 'strong' and d['config']['games_per_gpu'] == 262144 and d['config']['games_covered_by_ranks'] == 2097152
    assert d['config']['solo']['games'] == 262144 and d['config']['scaling_x'] > 0
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
    """65,536 games per GPU (BASELINE config 2) on any number of GPUs -- the same per-GPU workload, so that `value` over the GPU counts
    is one curve -- with config 5 (262,144 per GPU) and the strong split of 2,097,152 games as legs of a multi-GPU line, each with its
    solo anchor; the launcher parent counts GPUs from sysfs without importing torch."""
    p = _run(['--gpus', '1', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert p.returncode == 0, p.stderr
    d = _json_lines(p.stdout)[0]
    assert d['config']['games_per_gpu'] == 65536 and d['config']['strong_leg_total_games'] == 0 and d['config']['output_sets'] == 3
    assert d['config']['solo'] is None and d['config']['scaling_x'] is None and d['config']['scaling_legs'] is None
    p = _run(['--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert p.returncode == 0, p.stderr
    d = _json_lines(p.stdout)[0]
    c = d['config']
    assert c['games_per_gpu'] == 65536 and c['total_games'] == 131072 and c['output_sets'] == 3
    assert c['strong_leg_total_games'] == 2097152 and c['solo']['games'] == 65536 and c['scaling_x_ideal'] == 2
    assert [(l['games_per_gpu'], l['total_games'], l['scaling']) for l in c['scaling_legs']] == [(262144, 524288, 'weak'), (1048576, 2097152, 'strong')]
    assert all(l['solo']['games'] == l['games_per_gpu'] and l['scaling_x'] > 0 for l in c['scaling_legs'])
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpus(); "
            "assert isinstance(n, int) and n >= 0; assert 'torch' not in sys.modules; print('ok', n)" % ROOT)
    q = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert q.returncode == 0 and q.stdout.startswith('ok'), q.stderr


def test_only_rank_0_of_a_one_gpu_run_takes_the_single_gpu_legs():
    """legs_for: the CPU baseline, the live counter passes, the store probe, the facade, the other workloads and the trajectory leg run on
    rank 0 of a ONE-GPU run and nowhere else; every rank of a multi-GPU job runs the same (barrier-carrying) legs, so nobody waits at a
    barrier for a rank that is busy elsewhere."""
    import bench
    solo_only = {'cpu_baseline', 'live_traffic', 'store_probe', 'facade_n1', 'other_workloads', 'consumer_in_loop', 'trajectory', 'in_place', 'two_chains'}
    a1 = bench.parse_args(['--gpus', '1'])
    assert solo_only <= bench.legs_for(0, 1, a1) and 'scaling_legs' not in bench.legs_for(0, 1, a1)
    for world in (2, 4, 8):
        a = bench.parse_args(['--gpus', str(world)])
        per_rank = [bench.legs_for(r, world, a) for r in range(world)]
        assert all(l == per_rank[0] for l in per_rank), world            # the same legs on every rank
        assert not (per_rank[0] & solo_only) and 'scaling_legs' in per_rank[0] and {'no_settle', 'one_launch_per_step'} <= per_rank[0]
        assert bench.legs_for(3 % world, world, a, dry=True) == {'scaling_legs'}
    a = bench.parse_args(['--gpus', '1', '--no-cpu-baseline', '--no-other-workloads', '--no-live-traffic', '--no-store-probe', '--no-facade-leg'])
    assert not (bench.legs_for(0, 1, a) & {'cpu_baseline', 'live_traffic', 'store_probe', 'facade_n1', 'other_workloads', 'trajectory'})
