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


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'TORCHELASTIC_RUN_ID', 'SGX_BENCH_LAUNCHER')}
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return lines[0]


COMMON = ['--steps', '24', '--warmup', '6', '--no-cpu-baseline', '--no-other-workloads', '--verify-envs', '16', '--placement', 'plain',
          '--wake-seconds', '0']


@pytest.mark.parametrize('version,per_rank', [('barrage', 4096), ('micro', 6000)])
def test_two_ranks_on_one_gpu_equal_one_rank_and_the_oracle(version, per_rank):
    two = _run(['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--envs', str(per_rank), '--version', version] + COMMON)
    one = _run(['--gpus', '1', '--envs', str(2 * per_rank), '--version', version] + COMMON)
    assert two['n_gpus'] == 2 and two['config']['launched_by'] == 'bench.py' and two['config']['reduction_backend'] == 'gloo'
    assert two['config']['total_games'] == one['config']['total_games'] == 2 * per_rank
    assert two['config']['games_per_gpu'] == per_rank and two['scaling'] == 'weak'
    assert two['verified_envs'] >= 32 and one['verified_envs'] >= 16          # both ranks checked their sample against the oracle
    assert two['verified_steps'] == one['verified_steps'] == 30
    # one GPU: the headline writes a ring of three output sets (DRAM-side figures); several GPUs: in place.  Same games either way.
    assert one['config']['output_sets'] == 3 and two['config']['output_sets'] == 1
    assert two['config']['outputs_checksum'] == one['config']['outputs_checksum']
    assert two['value'] > 0 and two['config']['per_gpu_value_min'] <= two['config']['per_gpu_value_max']
    # the 1-rank run then played the in-place leg and its two-chains variant on the same env object, each verified against the oracle
    inp = one['config']['in_place']
    assert inp['verified_steps'] == 60 and inp['verified_envs'] >= 8 and inp['rate_over_spec_peak'] > 0
    assert one['config']['two_chains']['verified_steps'] == 90 and one['config']['two_chains']['verified_envs'] >= 16
    assert one['roofline']['frac_dram'] == one['roofline']['frac'] and one['roofline']['in_place_rate_over_spec_peak'] == inp['rate_over_spec_peak']
    assert two['roofline']['frac_dram'] is None and len(one['build_id']) == 16


def test_rccl_refusing_the_job_falls_back_to_gloo():
    """The default backend of the reporting reductions is nccl (= RCCL).  Two ranks on ONE GPU are something RCCL refuses ("invalid usage"):
    the real failure path -- every rank falls back to gloo on the host, the run completes with the same games, and the line says what
    happened.  (On a node with a GPU per rank the same code comes up on RCCL; that is the driver's to run.)"""
    two = _run(['--gpus', '2', '--devices', '0,0', '--envs', '4096'] + COMMON)
    ref = _run(['--gpus', '2', '--devices', '0,0', '--backend', 'gloo', '--envs', '4096'] + COMMON)
    assert two['n_gpus'] == 2 and two['config']['reduction_backend'].startswith('gloo (nccl failed: ')
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
