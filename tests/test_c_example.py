"""examples/c_rollout.c: the C ABI used from plain C (gcc, no Python / torch in the process).
CPU: it compiles against include/stratego_mi355x.h as C11 and links.  GPU: its env-0 digest equals the oracle's."""
import os
import re
import subprocess

import pytest

from stratego_env_amd import build as hip_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'examples', 'c_rollout.c')
EXE = os.path.join(ROOT, 'examples', '_build', 'c_rollout')
TABLE = os.path.join(ROOT, 'stratego_env_amd', 'inits', 'barrage_setups.npy')


def compile_example():
    hip_build.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.dirname(hip_build.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c11', '-O2', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                           '-I', os.path.join(ROOT, 'include'), SRC, '-L', libdir, '-lstratego_mi355x', '-L/opt/rocm/lib',
                           '-lamdhip64', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-o', EXE])
    return EXE


def test_header_is_plain_c_and_example_links():
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-fsyntax-only', '-x', 'c',
                           os.path.join(ROOT, 'include', 'stratego_mi355x.h')])
    assert os.path.exists(compile_example())


@pytest.mark.gpu
def test_c_rollout_matches_oracle():
    from oracle import oracle as orc
    from stratego_env_amd import setups as S
    from stratego_env_amd.config import VARIANTS
    exe = compile_example()
    n_envs, steps, seed = 1024, 200, 0xC0FFEE
    out = subprocess.check_output([exe, TABLE, str(n_envs), str(steps), hex(seed)], text=True)
    m = re.search(r'games_finished (\d+) invalid_actions (\d+) env0_digest 0x([0-9a-f]+)', out)
    assert m, out
    assert int(m.group(2)) == 0
    v = VARIANTS['barrage']
    cv = orc.make_cvariant(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts, v.initial_state_usable_rows,
                           setups=S.load_setup_table('barrage'))
    _, d, _ = orc.rollout(cv, seed, 0, 1, steps, threads=1)
    assert int(d[0]) == int(m.group(3), 16)


@pytest.mark.gpu
def test_c_rollout_trajectory_mode_matches_oracle_and_the_step_by_step_mode():
    """`c_rollout <table> <envs> <steps> <seed> traj`: the same rollout as ONE sgx_step_traj call from plain C into [steps][envs]... tensors
    (per-slot results, drawn actions); the digest over env 0's slots equals the oracle's rolling digest, i.e. the step-by-step mode's, and
    the steps ran as a multi-step launch (SGX_LAUNCH_MULTI_STEP_WAVE = 3)."""
    from oracle import oracle as orc
    from stratego_env_amd import setups as S
    from stratego_env_amd.config import VARIANTS
    exe = compile_example()
    n_envs, steps, seed = 777, 120, 0xBADC0DE
    out = subprocess.check_output([exe, TABLE, str(n_envs), str(steps), hex(seed), 'traj'], text=True)
    m = re.search(r'trajectory of 120 slots: .* games_finished (\d+) invalid_actions (\d+) env0_digest 0x([0-9a-f]+) launch_kind (\d+)', out)
    assert m and int(m.group(2)) == 0 and int(m.group(4)) == 3, out
    v = VARIANTS['barrage']
    cv = orc.make_cvariant(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts, v.initial_state_usable_rows,
                           setups=S.load_setup_table('barrage'))
    _, d, _ = orc.rollout(cv, seed, 0, 1, steps, threads=1)
    assert int(d[0]) == int(m.group(3), 16)
    step_by_step = subprocess.check_output([exe, TABLE, str(n_envs), str(steps), hex(seed)], text=True)
    assert re.search(r'env0_digest 0x([0-9a-f]+)', step_by_step).group(1) == m.group(3)


@pytest.mark.gpu
def test_c_rollout_bench_mode_uses_library_owned_outputs():
    """`c_rollout <table> <envs> <steps> <seed> bench`: output buffers from sgx_alloc_outputs (bounded placement trial), K steps
    through sgx_step_n -- plain C, no torch in the process."""
    exe = compile_example()
    out = subprocess.check_output([exe, TABLE, '8192', '64', '0x1', 'bench'], text=True)
    m = re.search(r'placement trial: (\d+) candidates, first ([0-9.]+) us, kept ([0-9.]+) us, peak extra ([0-9.]+) GiB', out)
    assert m and int(m.group(1)) >= 2 and float(m.group(3)) <= float(m.group(2)) and float(m.group(4)) <= 8.0, out
    assert re.search(r'envs 8192 steps 64: [0-9.]+ us per batched step, [0-9.]+ M env steps/s', out), out
