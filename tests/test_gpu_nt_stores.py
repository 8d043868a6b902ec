"""The non-temporal-store path of the observation writes (sgx_obs.h: emit_codes<..., NT = true>, odd boards, shared-wave boards,
patch_uncoded after an NT bulk store, BOTH mode, concurrent chains) against the oracle.

The library turns NT stores on by itself only when a launch writes more than 300 MB of observations, so the ordinary parity
suites (tens of envs) never reach it.  Here (a) the same suites run with the path forced on (SGX_NT=1 / sgx_set_nt_stores) for
all 10 variant configs and the 5 custom geometries, in partial, BOTH and 'original' mode, and (b) every configuration with a
quoted full-size figure is run at that size -- where the library chooses NT itself -- with the oracle's
digests of sampled envs."""
import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd import config
from stratego_env_amd import setups as S
from stratego_env_amd.config import VARIANTS
from tests.helpers import oracle_cvariant
from tests.test_gpu_generic_geometry import CUSTOM

pytestmark = pytest.mark.gpu


@pytest.fixture
def nt_forced(monkeypatch):
    monkeypatch.setenv('SGX_NT', '1')        # read by sgx_create: every handle of the test gets the NT path
    monkeypatch.setenv('SGX_XCD_SKEW', '150')  # ... and the unequal XCD shares the library uses for streaming launches (group_of_block)


@pytest.fixture
def custom_names():
    config.VARIANTS.update(CUSTOM)
    yield
    for k in CUSTOM:
        config.VARIANTS.pop(k, None)


def _table(name):
    v = VARIANTS[name]
    return S.load_setup_table(v.human_inits) if v.human_inits else None


@pytest.mark.parametrize('name,n_envs,n_steps,garbage', [
    ('barrage', 32, 400, 0.1), ('standard', 16, 500, 0.05), ('octa_barrage', 32, 300, 0.1), ('medium', 32, 250, 0.1),
    ('fives', 32, 200, 0.1), ('tiny', 64, 200, 0.1), ('micro', 64, 120, 0.1), ('short_barrage', 32, 200, 0.1),
    ('short_standard', 8, 200, 0.05), ('standard2', 4, 250, 0.05),
])
def test_step_parity_with_nt_forced(nt_forced, name, n_envs, n_steps, garbage):
    """Every output of every step (and the terminal observations) with all whole lines leaving as non-temporal stores.
    standard / standard2 run long enough for captured miners / majors / bombs / colonels: values without a 4-bit code, written by
    patch_uncoded (4-aligned boards) / patch_uncoded_floats after s_waitcnt (odd boards) next to NT bulk stores."""
    from tests.test_gpu_parity import test_step_bit_exact_vs_oracle
    test_step_bit_exact_vs_oracle(name, n_envs, n_steps, garbage, seed_salt=3)


@pytest.mark.parametrize('name,n_envs,n_steps', [('c3x3', 48, 80), ('c7x7', 32, 200), ('c9x5', 32, 200), ('c12x12', 12, 250), ('c3x40', 12, 120),
                                                 ('c20x20', 4, 200)])
def test_custom_geometries_with_nt_forced(nt_forced, custom_names, name, n_envs, n_steps):
    from tests.test_gpu_parity import test_step_bit_exact_vs_oracle
    test_step_bit_exact_vs_oracle(name, n_envs, n_steps, 0.1, seed_salt=4)


@pytest.mark.parametrize('name,n_envs,n_steps', [('barrage', 32, 400), ('standard', 6, 400), ('tiny', 48, 150), ('micro', 48, 80),
                                                 ('fives', 32, 150), ('octa_barrage', 16, 200), ('medium', 16, 150), ('standard2', 2, 200)])
def test_both_observations_with_nt_forced(nt_forced, name, n_envs, n_steps):
    from tests.test_gpu_full_obs import check_both_obs_vs_oracle
    check_both_obs_vs_oracle(name, n_envs, n_steps, 'extended')


@pytest.mark.parametrize('name,n_envs,n_steps', [('barrage', 32, 300), ('standard', 6, 200), ('micro', 48, 80), ('fives', 32, 150),
                                                 ('standard2', 2, 60)])
def test_original_channels_with_nt_forced(nt_forced, name, n_envs, n_steps):
    """(the 'original' kinds keep the LUT emission with plain stores: the switch must leave them alone)"""
    from tests.test_gpu_full_obs import check_both_obs_vs_oracle
    check_both_obs_vs_oracle(name, n_envs, n_steps, 'original')


def test_functional_api_with_nt_forced(nt_forced):
    from tests.test_gpu_procedural import test_functional_api_matches_oracle
    for name in ('barrage', 'fives'):
        test_functional_api_matches_oracle(name)


@pytest.mark.parametrize('name,n', [('barrage', 3000), ('barrage', 1), ('barrage', 9), ('fives', 2001), ('micro', 5003), ('standard2', 300)])
def test_switching_the_store_policy_changes_no_byte(name, n):
    """One env played under sgx_set_nt_stores 0 / 1 / auto in turn equals a second one left on auto."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=11, auto_reset=True, full_obs=True)
    b = VecStrategoEnv(name, n, seed=11, auto_reset=True, full_obs=True)
    a.reset(); b.reset()
    for k, mode in enumerate((True, False, 'auto', True)):
        a.set_nt_stores(mode)
        a.set_xcd_skew((0, 100, 333, 900)[k])             # which workgroup plays which game changes nothing either
        a.rollout_steps(23 + k)
        b.rollout_steps(23 + k)
        torch.cuda.synchronize()
        for x, y in ((a.obs, b.obs), (a.fobs, b.fobs), (a.mask, b.mask), (a.next_actions, b.next_actions)):
            assert torch.equal(x, y), (name, mode)
    with pytest.raises(Exception):
        a.set_nt_stores(7)
    a.close(); b.close()


def _digest_rows(env, idx, digs, both=False):
    mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
    fo = env.fobs[idx].cpu().numpy() if both else None
    rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
    ei = env.ending_invalid[idx].cpu().numpy()
    for i in range(len(digs)):
        digs[i] = orc.step_digest(mk[i], ob[i], rw[i], dn[i], pl[i], ei[i], fobs=None if fo is None else fo[i], h=digs[i])


def _full_size_check(name, N, warm, T, n_chk, both=False, chains=1, seed=0x5712A7E60):
    """N games: `warm` untested rollout steps (one library call), then T steps whose outputs are digested for sampled envs and
    compared with the oracle's digest over the same steps of the same envs; final packed states against the oracle's."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[name]
    env = VecStrategoEnv(name, N, seed=seed, auto_reset=True, full_obs=both)
    env.reset()
    if warm:
        env.rollout_steps(warm, chains=chains)
    ids = np.unique(np.concatenate([np.arange(0, 5), np.arange(N - 5, N), np.linspace(0, N - 1, n_chk).astype(np.int64)]))
    idx = torch.from_numpy(ids).to(env.device)
    digs = [orc.FNV_OFFSET] * len(ids)
    for t in range(T):
        if chains > 1:
            env.rollout_steps(1, chains=chains)
        else:
            env.rollout_step()
        _digest_rows(env, idx, digs, both)
    assert int(env.invalid_action.sum()) == 0
    m = env.mask.view(N, -1)
    assert int((m.sum(dim=1, dtype=torch.int32) == 0).sum()) == 0 and int((m > 1).sum()) == 0
    assert bool(torch.isfinite(env.obs).all()) and float(env.obs.abs().max()) <= 1.0
    st, pl = env.export_state()
    st_h, info_h = st[idx].cpu().numpy(), env.env_info()[idx].cpu().numpy()
    cv = oracle_cvariant(name, setups=_table(name))
    for i, g in enumerate(ids):
        r = orc.rollout_ex(cv, seed, int(g), 1, warm + T, skip=warm, both=both, want_states=True)
        assert int(r['digests'][0]) == digs[i], (name, 'env', int(g))
        assert np.array_equal(r['states'][0], st_h[i]), (name, 'env', int(g), 'final state')
        assert np.array_equal(r['info'][0], info_h[i]), (name, 'env', int(g), 'turn / game / over / player')
    env.close()
    del env
    torch.cuda.empty_cache()


def test_standard2_full_size_odd_board_nt():
    """standard2 (15x15, colonel x3 => thirds: uncoded entries) at the size of its quoted figure: 32,768 games = 1.98 GB of
    observations per launch, NT chosen by the library; 260 warm-up steps so that captures have happened."""
    _full_size_check('standard2', 32768, 260, 12, 6)


@pytest.mark.parametrize('name', ['micro', 'tiny'])
def test_toy_boards_262144_games_nt(name):
    """The shared-wave boards (4 games per wave) where their 1.81 G steps/s figure is quoted: 262,144 games, NT by size."""
    _full_size_check(name, 262144, 40, 24, 16)


def test_fives_and_medium_octa_at_nt_size():
    """5x5 (two games per wave, odd board), 6x6 and 8x8 past the Infinity Cache size."""
    _full_size_check('fives', 65536 * 4, 30, 16, 10)
    _full_size_check('medium', 65536 * 2, 60, 12, 8)
    _full_size_check('octa_barrage', 65536, 200, 12, 8)


def test_barrage_both_mode_full_size_nt():
    """The reference's default observation mode (BOTH_OBSERVATIONS) at 65,536 Barrage games: both observations digested."""
    _full_size_check('barrage', 65536, 150, 16, 12, both=True)


def test_standard_262144_mid_game_nt_with_uncoded_entries():
    """BASELINE config 3 size, 220 steps into the games (captured miners / majors / bombs have fifths and thirds as normalised
    counts: emit_codes<CHECKED, NT> + patch_uncoded on most envs)."""
    _full_size_check('standard', 262144, 220, 8, 10)


def test_barrage_262144_games_config5_per_gpu_size():
    """BASELINE config 5's per-GPU share (262,144 Barrage games; `bench.py --gpus N > 1` runs this size on every rank): oracle digests of
    sampled envs and their final states, 150 steps into the games."""
    _full_size_check('barrage', 262144, 150, 8, 12)


def test_standard_both_mode_mid_game_nt_mixed_lines():
    """Standard in BOTH_OBSERVATIONS mode past the Infinity Cache size, 220 steps into the games: both renderings take
    emit_codes<CHECKED, NT>, where the lines that hold a left-out quad leave as plain stores and are completed by patch_uncoded;
    and 6x6 Standard (medium_standard), whose lines are shared between neighbouring games more often."""
    _full_size_check('standard', 32768, 220, 8, 10, both=True)
    _full_size_check('medium_standard', 131072, 120, 8, 8)


def test_two_chains_full_size_equal_one_chain_and_the_oracle():
    """sgx_rollout(chains = 2) at 65,536 Barrage games (NT by size, the two ranges on streams of their own): oracle digests of
    sampled envs from both ranges, and the final state equals a single-chain run's."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    _full_size_check('barrage', 65536, 100, 10, 12, chains=2)
    N, seed = 65536, 0xC4A1
    a = VecStrategoEnv('barrage', N, seed=seed, auto_reset=True)
    b = VecStrategoEnv('barrage', N, seed=seed, auto_reset=True)
    a.reset(); b.reset()
    a.rollout_steps(64, chains=2)
    b.rollout_steps(64)
    torch.cuda.synchronize()
    assert torch.equal(a.obs, b.obs) and torch.equal(a.mask, b.mask) and torch.equal(a.next_actions, b.next_actions)
    sa, pa = a.export_state(); sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    a.close(); b.close()
