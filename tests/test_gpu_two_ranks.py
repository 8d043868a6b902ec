"""The multi-rank PRODUCT path on hardware: bench.py's own launcher starts two ranks that share the one GPU of the test box
(--devices 0,0; the reporting reductions go over gloo because RCCL refuses two ranks on one device), each rank runs the real
Rank / VecStrategoEnv / sharding / reduce code on its contiguous range of global env ids, verifies sampled envs against the CPU
oracle, and the checksum of checksums over all envs must equal a single-rank run's over the same global ids (trajectories do not
depend on the sharding: every draw is keyed by the global env id)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, timeout=600, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'TORCHELASTIC_RUN_ID', 'SGX_BENCH_LAUNCHER')}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return lines[0]


COMMON = ['--steps', '24', '--warmup', '6', '--no-cpu-baseline', '--no-other-workloads', '--verify-envs', '16', '--placement', 'plain',
          '--wake-seconds', '0', '--no-live-traffic', '--no-store-probe', '--no-facade-leg']


@pytest.mark.parametrize('version,per_rank', [('barrage', 4096), ('micro', 6000)])
def test_two_ranks_on_one_gpu_equal_one_rank_and_the_oracle(version, per_rank):
    two = _run(['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--envs', str(per_rank), '--version', version] + COMMON)
    one = _run(['--gpus', '1', '--envs', str(2 * per_rank), '--version', version] + COMMON)
    assert two['n_gpus'] == 2 and two['config']['launched_by'] == 'bench.py' and two['config']['reduction_backend'] == 'gloo'
    assert two['config']['total_games'] == one['config']['total_games'] == 2 * per_rank
    assert two['config']['games_per_gpu'] == per_rank and two['scaling'] == 'weak'
    assert two['verified_envs'] >= 32 and one['verified_envs'] >= 16          # both ranks checked their sample against the oracle
    assert two['verified_steps'] == one['verified_steps'] == 30
    # the headline writes a ring of three output sets (DRAM-side figures) on any number of GPUs: the same per-GPU workload, same games
    assert one['config']['output_sets'] == 3 and two['config']['output_sets'] == 3
    # rank 0 timed its 'per_rank' games ALONE before the two ranks ran side by side, and put the games back afterwards: the checksum of
    # checksums still equals the 1-rank run's (which has no anchor: it is its own)
    assert two['config']['outputs_checksum'] == one['config']['outputs_checksum']
    solo = two['config']['solo']
    assert solo['games'] == per_rank and solo['value'] > 0 and two['config']['scaling_x'] == pytest.approx(two['value'] / solo['value'])
    assert two['config']['scaling_x_ideal'] == 2 and two['config']['per_gpu_value_min_over_solo'] > 0
    assert one['config']['solo'] is None and one['config']['scaling_x'] is None
    assert two['value'] > 0 and two['config']['per_gpu_value_min'] <= two['config']['per_gpu_value_max']
    # both runs then played the headline's steps once more without gpu_settle (config.no_settle); the 1-rank run went on with the
    # in-place leg and its two-chains variant on the same env object, each verified against the oracle
    assert one['config']['no_settle']['launch_us'] > 0 and two['config']['no_settle']['value'] > 0
    inp = one['config']['in_place']
    # (headline 30 + the same without the settle 30 + the same as one launch per step 30, then the in-place leg's 30)
    assert one['config']['one_launch_per_step']['launch_us'] > 0 and one['config']['steps_per_launch'] == 24
    assert inp['verified_steps'] == 120 and inp['verified_envs'] >= 8 and inp['rate_over_spec_peak'] > 0
    # (... the in-place leg once more as one launch per step: 30, then the two chains' 30)
    assert inp['one_launch_per_step']['launch_us'] > 0 and inp['one_launch_per_step']['value'] > 0
    assert one['config']['two_chains']['verified_steps'] == 180 and one['config']['two_chains']['verified_envs'] >= 16
    assert one['roofline']['frac_dram'] == one['roofline']['frac'] and one['roofline']['in_place_rate_over_spec_peak'] == inp['rate_over_spec_peak']
    assert two['roofline']['frac_dram'] == two['roofline']['frac'] and len(one['build_id']) == 16


def test_scaling_legs_are_self_anchoring_on_two_ranks():
    """A multi-GPU line's legs (BASELINE config 5's games per GPU and the strong split; small stand-ins here: two ranks share the test
    box's one GPU) each carry their solo anchor -- rank 0 alone on the leg's per-GPU size, the other rank parked -- and scaling_x; the
    per-GPU workload of the HEADLINE is the 1-GPU line's (ring of three output sets)."""
    two = _run(['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--envs', '4096', '--leg-envs', '8192', '--strong-total', '6000'] + COMMON)
    c = two['config']
    assert c['output_sets'] == 3 and c['solo']['games'] == 4096 and c['scaling_x'] > 0
    legs = c['scaling_legs']
    assert [(l['games_per_gpu'], l['total_games'], l['scaling'], l['output_sets']) for l in legs] == [(8192, 16384, 'weak', 1), (3000, 6000, 'strong', 1)]
    for l in legs:
        assert l['solo']['games'] == l['games_per_gpu'] and l['solo']['launch_us'] > 0
        assert l['scaling_x'] == pytest.approx(l['value'] / l['solo']['value']) and l['scaling_x_ideal'] == 2
        assert l['per_gpu_value_min_over_solo'] > 0 and l['verified_envs'] >= 16


def test_rccl_refusing_the_job_falls_back_to_gloo():
    """The default backend of the reporting reductions is nccl (= RCCL).  Two ranks on ONE GPU are something RCCL refuses ("invalid usage"):
    the real failure path -- every rank falls back to gloo on the host, the run completes with the same games, and the line says what
    happened.  (On a node with a GPU per rank the same code comes up on RCCL; that is the driver's to run.)"""
    ref = _run(['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--envs', '4096'] + COMMON)
    # (a) class Rank's own check before the collective bring-up: a device shared by two ranks
    two = _run(['--gpus', '2', '--devices', '0,0', '--envs', '4096'] + COMMON)
    assert two['n_gpus'] == 2 and two['config']['reduction_backend'].startswith('gloo (nccl failed: rank 0: ')
    assert 'shared by 2 ranks' in two['config']['reduction_backend']
    assert two['config']['outputs_checksum'] == ref['config']['outputs_checksum'] and two['verified_envs'] >= 32 and two['value'] > 0
    # (b) that check switched off: RCCL itself refuses inside the collective bring-up, on every rank; the ranks agree on gloo afterwards
    import time
    t0 = time.perf_counter()
    two = _run(['--gpus', '2', '--devices', '0,0', '--envs', '4096'] + COMMON, extra_env={'SGX_BENCH_SKIP_DEVICE_CHECK': '1', 'SGX_BENCH_NCCL_TIMEOUT': '20'})
    # a first-time RCCL bring-up that cannot work ends in the gloo fallback within its timeout: the bring-up itself (both stages, the
    # agreement) and the whole job stay far from the minutes a hung collective would take
    assert two['config']['reduction_bringup_s'] < 60 and time.perf_counter() - t0 < 240
    assert two['n_gpus'] == 2 and two['config']['reduction_backend'].startswith('gloo (nccl failed: rank 0: ')
    assert 'shared by 2 ranks' not in two['config']['reduction_backend']
    assert two['config']['outputs_checksum'] == ref['config']['outputs_checksum'] and two['verified_envs'] >= 32 and two['value'] > 0


def test_eight_ranks_on_one_gpu_equal_one_rank_and_the_oracle():
    """World size 8 -- BASELINE config 5's layout -- on the one GPU of the test box: 8 x 8,192 Barrage games, every rank its own
    VecStrategoEnv on its range of global env ids, gloo for the reporting reductions; the checksum of checksums over all 65,536 games
    equals the 1-rank run's and every rank verified its sample against the CPU oracle."""
    eight = _run(['--gpus', '8', '--devices', '0,0,0,0,0,0,0,0', '--backend', 'gloo', '--envs', '8192', '--version', 'barrage',
                  '--output-sets', '1'] + COMMON, timeout=1500)
    one = _run(['--gpus', '1', '--envs', '65536', '--version', 'barrage', '--no-two-chains', '--output-sets', '1'] + COMMON, timeout=900)
    assert eight['n_gpus'] == 8 and eight['config']['launched_by'] == 'bench.py' and eight['config']['reduction_backend'] == 'gloo'
    assert eight['config']['total_games'] == one['config']['total_games'] == 65536 and eight['config']['games_per_gpu'] == 8192
    assert eight['verified_envs'] >= 8 * 16 and eight['verified_steps'] == 30
    assert eight['config']['outputs_checksum'] == one['config']['outputs_checksum']
    assert eight['config']['per_gpu_value_min'] <= eight['config']['per_gpu_value_max']


def test_strong_split_with_a_remainder_on_one_gpu():
    """--total-envs 8191 over 2 ranks (4096 + 4095 games): same checksum as one rank with 8191 games."""
    two = _run(['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--total-envs', '8191'] + COMMON)
    one = _run(['--gpus', '1', '--total-envs', '8191'] + COMMON)
    assert two['scaling'] == 'strong' and two['config']['total_games'] == 8191
    assert two['config']['outputs_checksum'] == one['config']['outputs_checksum']
    assert two['verified_envs'] >= 32


def test_under_torch_distributed_run_on_the_gpu():
    """The driver's launch form on hardware: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 (ranks from the
    environment; the two ranks share the test box's one GPU): one line, launched_by external, the anchors and the checksum of a
    self-launched run of the same games."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'TORCHELASTIC_RUN_ID', 'SGX_BENCH_LAUNCHER')}
    args = ['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--envs', '4096'] + COMMON
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), BENCH] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    ext = lines[0]
    own = _run(args)
    assert ext['n_gpus'] == 2 and ext['config']['launched_by'] == 'external' and own['config']['launched_by'] == 'bench.py'
    assert ext['config']['outputs_checksum'] == own['config']['outputs_checksum'] and ext['verified_envs'] >= 32
    assert ext['config']['solo']['games'] == 4096 and ext['config']['scaling_x'] > 0


def test_live_counter_traffic_of_the_headline():
    """roofline.traffic is measured in the run itself: two short children of bench.py under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE), the
    guide's gfx950 correction; it must agree with the byte minimum of the launch (the kernel moves its minimum and little else) and the
    static entry of profiles/traffic.json stays next to it."""
    line = _run(['--gpus', '1', '--envs', '16384', '--steps', '12', '--warmup', '4', '--no-cpu-baseline', '--no-other-workloads', '--no-two-chains',
                 '--no-in-place-leg', '--no-settle-leg', '--verify-envs', '8', '--placement', 'plain', '--wake-seconds', '0'], timeout=900)
    rf = line['roofline']
    assert 'traffic_live_failed' not in rf, rf.get('traffic_live_failed')
    assert rf['traffic_source'].startswith('live: rocprofv3') and rf['traffic_static_source'] is None or rf['traffic_static_source'].startswith('static')
    assert 0.97 < rf['traffic_over_b_min'] < 1.06, rf['traffic_over_b_min']
    assert abs(rf['traffic'] / rf['bytes_per_launch'] - rf['traffic_over_b_min']) < 1e-9
