"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle, bit-exact.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd import setups as S
from stratego_env_amd.config import VARIANTS
from tests.helpers import digest_obs, load_games, oracle_cvariant, oracle_env

pytestmark = pytest.mark.gpu

MASK, POBS = 'valid_actions_mask', 'partial_observation'


def _table(name):
    v = VARIANTS[name]
    return S.load_setup_table(v.human_inits) if v.human_inits else None


def _oracle_batch(name, seed, g0, n):
    cv = oracle_cvariant(name, setups=_table(name))
    envs = []
    for e in range(n):
        oe = oracle_env(name)
        oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, 0))
        oe.game_no = 0
        envs.append(oe)
    return cv, envs


@pytest.mark.parametrize('name,n_envs,n_steps,garbage', [
    ('barrage', 48, 700, 0.0), ('barrage', 32, 300, 0.15), ('standard', 16, 500, 0.05), ('octa_barrage', 32, 300, 0.1),
    ('medium', 32, 300, 0.1), ('fives', 32, 200, 0.1), ('tiny', 64, 200, 0.1), ('micro', 64, 120, 0.1),
    ('short_barrage', 32, 250, 0.1), ('standard2', 4, 150, 0.05),
])
def test_step_bit_exact_vs_oracle(name, n_envs, n_steps, garbage, seed_salt=0, require_endings=True, final_obs=True, lane_kernel='auto', emit_obs=True):
    """Random-valid-action rollouts with auto-reset (+ injected garbage actions): every output of every step.
    (tools/soak_parity.py re-runs this with other seeds, batch sizes and garbage rates for minutes.)
    emit_obs=False (with final_obs=False): the steps pass no observation pointer -- the no-observation kernel kind -- and everything
    else (mask, rewards, flags, sampled action, exported state) is compared as before."""
    assert emit_obs or not final_obs
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[name]
    seed, g0 = 0xABCDEF12345 + len(name) + 7919 * seed_salt, 1000 + 31 * seed_salt
    env = VecStrategoEnv(name, n_envs, seed=seed, env_id_offset=g0, auto_reset=True, final_obs=final_obs)
    env.set_lane_kernel(lane_kernel)
    cv, oenvs = _oracle_batch(name, seed, g0, n_envs)
    NA = v.num_spatial_actions
    rs = np.random.RandomState(7 + seed_salt)
    obs, mask, player = env.reset()
    obs_h, mask_h, player_h = obs.cpu().numpy(), mask.cpu().numpy(), player.cpu().numpy()
    cur = []
    for e, oe in enumerate(oenvs):
        o = oe._obs(1)
        assert np.array_equal(o[MASK], mask_h[e]) and o[POBS].tobytes() == obs_h[e].tobytes(), (name, 'reset', e)
        assert player_h[e] == 1
        cur.append(o)
    st, pl = env.export_state()
    assert np.array_equal(st.cpu().numpy(), np.stack([oe.state for oe in oenvs]))
    sampled = env.sample_valid_actions().cpu().numpy()
    games_done = 0
    for t in range(n_steps):
        acts = np.zeros(n_envs, dtype=np.int32)
        for e, oe in enumerate(oenvs):
            a = orc.sample_action(cur[e][MASK].astype(np.uint8), seed, g0 + e, oe.game_no, int(oe.state[5, 0, 0]))
            assert sampled[e] == a, (name, t, e, 'sampler')
            if garbage and rs.rand() < garbage:
                a = int(rs.choice([rs.randint(NA), rs.randint(v.cells) * v.spatial_channels + v.spatial_channels - 1,
                                   -1, NA, NA + 3, (v.cells - 1) * v.spatial_channels + rs.randint(v.spatial_channels)]))
            acts[e] = a
        env.step(torch.from_numpy(acts), want_next_actions=True, emit_obs=emit_obs)
        obs_h, mask_h = env.obs.cpu().numpy(), env.mask.cpu().numpy()
        rew_h, done_h, player_h = env.reward.cpu().numpy(), env.done.cpu().numpy(), env.player.cpu().numpy()
        inv_h, einv_h = env.invalid_action.cpu().numpy(), env.ending_invalid.cpu().numpy()
        fin_h = env.final_obs.cpu().numpy() if final_obs else None
        sampled = env.next_actions.cpu().numpy()
        for e, oe in enumerate(oenvs):
            try:
                o, rew, done, info = oe.step({oe.player: int(acts[e])})
            except ValueError:
                assert inv_h[e] == 1, (name, t, e, acts[e], 'oracle raised, GPU accepted')
                assert not done_h[e]
                assert np.array_equal(cur[e][MASK], mask_h[e]) and (not emit_obs or cur[e][POBS].tobytes() == obs_h[e].tobytes())
                assert player_h[e] == oe.player
                continue
            assert inv_h[e] == 0, (name, t, e, acts[e], 'GPU flagged a move the oracle accepts')
            assert bool(done_h[e]) == done['__all__'], (name, t, e)
            if done['__all__']:
                games_done += 1
                assert (rew_h[e, 0], rew_h[e, 1]) == (rew[1], rew[-1]), (name, t, e, rew_h[e], rew)
                assert bool(einv_h[e]) == info[1]['game_result_was_invalid']
                for slot, p in ((0, 1), (1, -1)):
                    assert fin_h is None or o[p][POBS].tobytes() == fin_h[e, slot].tobytes(), (name, t, e, 'final obs', p)
                    assert int(o[p][MASK].sum()) == 1 and o[p][MASK][0, 0, -1] == 1
                oe.game_no += 1
                o = oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, oe.game_no))
            else:
                assert rew_h[e, 0] == 0 and rew_h[e, 1] == 0 and einv_h[e] == 0
            p = oe.player
            assert player_h[e] == p, (name, t, e)
            assert np.array_equal(o[p][MASK], mask_h[e]), (name, t, e, 'mask')
            assert not emit_obs or o[p][POBS].tobytes() == obs_h[e].tobytes(), (name, t, e, 'obs')
            cur[e] = o[p]
        if t % 50 == 49:
            st, pl = env.export_state()
            assert np.array_equal(st.cpu().numpy(), np.stack([oe.state for oe in oenvs])), (name, t, 'state export')
            assert np.array_equal(pl.cpu().numpy(), np.asarray([oe.player for oe in oenvs], dtype=np.int8))
    assert games_done > 0 or not require_endings or name in ('standard', 'standard2', 'medium_standard', 'short_standard', 'c12x12', 'c20x20', 'c17x16', 'c32x32')
    env.close()


@pytest.mark.parametrize('name,n_envs,n_steps,garbage', [('barrage', 40, 500, 0.1), ('standard', 12, 400, 0.05), ('octa_barrage', 24, 300, 0.1),
                                                        ('medium', 24, 250, 0.1), ('fives', 24, 200, 0.1), ('standard2', 3, 150, 0.05)])
def test_step_without_observation_bit_exact_vs_oracle(name, n_envs, n_steps, garbage):
    """The no-observation kernel kind (steps that pass no observation pointer: mask-only rollouts, search expansions) against the
    oracle, step by step: masks, rewards, flags, sampled actions, exported states.  (Toy boards: the lane kernel plays such launches,
    tests/test_gpu_lane_kernel.py; tests/test_gpu_no_obs_kind.py forces the wave-per-game kind there.)"""
    test_step_bit_exact_vs_oracle(name, n_envs, n_steps, garbage, seed_salt=3, final_obs=False, emit_obs=False)


def test_config2_256_heldout_seeds_to_termination():
    """BASELINE config 2's bit-exact check, literally: 256 held-out seeds bench_seed+1 .. bench_seed+256, env id 0,
    each played with the workload's own action rule (k-th valid action, k from the counter RNG) until its first game ends, on
    the GPU (one handle per seed -- the seed belongs to the handle -- writing into slices of shared output tensors) and on the
    CPU oracle; compared per step on (drawn action, mask, observation, reward, done, player, invalid-ending) and on the final
    state."""
    import torch
    from bench import BASE_SEED
    from stratego_env_amd.vec_env import VecStrategoEnv
    name, n = 'barrage', 256
    v = VARIANTS[name]
    cv = oracle_cvariant(name, setups=_table(name))
    dev = torch.device('cuda', 0)
    R, Cc, K = v.rows, v.columns, v.spatial_channels
    obs_all = torch.empty((n, R, Cc, 67), dtype=torch.float32, device=dev)
    mask_all = torch.empty((n, R, Cc, K), dtype=torch.uint8, device=dev)
    rew_all = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    done_all = torch.zeros((n,), dtype=torch.uint8, device=dev)
    player_all = torch.zeros((n,), dtype=torch.int8, device=dev)
    inv_all = torch.zeros((n,), dtype=torch.uint8, device=dev)
    einv_all = torch.zeros((n,), dtype=torch.uint8, device=dev)
    next_all = torch.zeros((n,), dtype=torch.int32, device=dev)
    envs, oenvs, cur = [], [], []
    for i in range(n):
        seed = BASE_SEED + 1 + i
        e = VecStrategoEnv(name, 1, seed=seed, env_id_offset=0, auto_reset=False)
        e.obs, e.mask, e.reward, e.done = obs_all[i:i + 1], mask_all[i:i + 1], rew_all[i:i + 1], done_all[i:i + 1]
        e.player, e.invalid_action, e.ending_invalid, e.next_actions = player_all[i:i + 1], inv_all[i:i + 1], einv_all[i:i + 1], next_all[i:i + 1]
        e.reset()
        e.sample_valid_actions()
        envs.append(e)
        oe = oracle_env(name)
        oe.reset(initial_state_override=orc.reset_state(cv, seed, 0, 0))
        oenvs.append(oe)
    obs_h, mask_h, nxt_h = obs_all.cpu().numpy(), mask_all.cpu().numpy(), next_all.cpu().numpy()
    for i, oe in enumerate(oenvs):
        o = oe._obs(1)
        assert np.array_equal(o[MASK], mask_h[i]) and o[POBS].tobytes() == obs_h[i].tobytes(), ('reset', i)
        cur.append(o)
    live = list(range(n))
    steps = 0
    while live:
        acts = {}
        for i in live:
            oe = oenvs[i]
            a = orc.sample_action(cur[i][MASK].astype(np.uint8), BASE_SEED + 1 + i, 0, 0, int(oe.state[5, 0, 0]))
            assert nxt_h[i] == a, (i, steps, 'drawn action')
            acts[i] = a
            envs[i].rollout_step()
        obs_h, mask_h, nxt_h = obs_all.cpu().numpy(), mask_all.cpu().numpy(), next_all.cpu().numpy()
        rew_h, done_h, player_h = rew_all.cpu().numpy(), done_all.cpu().numpy(), player_all.cpu().numpy()
        inv_h, einv_h = inv_all.cpu().numpy(), einv_all.cpu().numpy()
        still = []
        for i in live:
            oe = oenvs[i]
            o, rew, done, info = oe.step({oe.player: acts[i]})
            assert inv_h[i] == 0 and bool(done_h[i]) == done['__all__'], (i, steps)
            if done['__all__']:
                assert (rew_h[i, 0], rew_h[i, 1]) == (rew[1], rew[-1]) and bool(einv_h[i]) == info[1]['game_result_was_invalid']
                assert int(mask_h[i].sum()) == 1 and mask_h[i][0, 0, -1] == 1
                st, pl = envs[i].export_state()
                assert np.array_equal(st.cpu().numpy()[0], oe.state), (i, 'final state')
            else:
                p = oe.player
                assert player_h[i] == p and np.array_equal(o[p][MASK], mask_h[i]) and o[p][POBS].tobytes() == obs_h[i].tobytes(), (i, steps)
                assert rew_h[i, 0] == 0 and rew_h[i, 1] == 0
                cur[i] = o[p]
                still.append(i)
        live = still
        steps += 1
        assert steps <= v.max_turns
    for e in envs:
        e.close()


@pytest.mark.parametrize('name', ['barrage', 'standard', 'octa_barrage', 'medium', 'fives', 'tiny', 'micro', 'short_barrage',
                                  'short_standard', 'standard2'])
def test_replay_reference_golden_games(name, limit=1 << 30):
    """Golden games recorded from the REFERENCE (tests/golden, tools/oracle/gen_golden.py) -- ALL of them: 256 Barrage, 24
    Standard, every toy and short variant: the GPU must reproduce every per-step digest, reward, done flag, error flag and the
    final internal state."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    g = load_games(name)
    off = g['offsets']
    n = min(len(off) - 1, limit)
    env = VecStrategoEnv(name, n, seed=1, auto_reset=False, final_obs=True, human_inits=False)
    obs, mask, player = env.reset(torch.from_numpy(g['p1_maps'][:n]), torch.from_numpy(g['p2_maps'][:n]))
    obs_h, mask_h = obs.cpu().numpy(), mask.cpu().numpy()
    for e in range(n):
        assert digest_obs({1: {MASK: mask_h[e], POBS: obs_h[e]}}) == int(g['init_digests'][e])
    lens = off[1:n + 1] - off[:n]
    final_states = {}
    for t in range(int(lens.max())):
        acts = np.zeros(n, dtype=np.int32)
        live = [e for e in range(n) if t < lens[e]]
        for e in live:
            acts[e] = g['actions'][off[e] + t]
        # finished envs keep receiving an out-of-range action: flagged invalid, state untouched
        for e in range(n):
            if t >= lens[e]:
                acts[e] = -1
        env.step(torch.from_numpy(acts))
        obs_h, mask_h = env.obs.cpu().numpy(), env.mask.cpu().numpy()
        rew_h, done_h, player_h = env.reward.cpu().numpy(), env.done.cpu().numpy(), env.player.cpu().numpy()
        inv_h, fin_h = env.invalid_action.cpu().numpy(), env.final_obs.cpu().numpy()
        for e in live:
            k = off[e] + t
            assert bool(inv_h[e]) == bool(g['errors'][k]), (name, e, t, 'error flag')
            if g['errors'][k]:
                continue
            assert bool(done_h[e]) == bool(g['dones'][k]) and player_h[e] == g['players'][k], (name, e, t)
            if done_h[e]:
                noop = np.zeros_like(mask_h[e]); noop[0, 0, -1] = 1
                d = digest_obs({1: {MASK: noop, POBS: fin_h[e, 0]}, -1: {MASK: noop, POBS: fin_h[e, 1]}})
                assert np.array_equal(mask_h[e], noop)
                assert tuple(rew_h[e]) == tuple(g['rewards'][k])
            else:
                d = digest_obs({int(player_h[e]): {MASK: mask_h[e], POBS: obs_h[e]}})
            assert d == int(g['digests'][k]), (name, e, t, 'digest')
    st, pl = env.export_state()
    st = st.cpu().numpy()
    for e in range(n):
        assert np.array_equal(st[e], g['final_states'][e].astype(np.int64)), (name, e, 'final state')
    env.close()


def test_import_export_roundtrip_and_observe():
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    name = 'barrage'
    g = load_games(name)
    n = 32
    states = g['final_states'][:n].astype(np.int64)
    env = VecStrategoEnv(name, n, seed=3)
    players = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int8)
    obs, mask, pl = env.import_state(torch.from_numpy(states), torch.from_numpy(players))
    st, pl2 = env.export_state()
    assert np.array_equal(st.cpu().numpy(), states) and np.array_equal(pl2.cpu().numpy(), players)
    oe = oracle_env(name)
    obs_h, mask_h = obs.cpu().numpy(), mask.cpu().numpy()
    for e in range(n):
        oe.reset(initial_state_override=states[e], first_player_override=int(players[e]))
        o = oe._obs(int(players[e]))
        assert np.array_equal(o[MASK], mask_h[e]) and o[POBS].tobytes() == obs_h[e].tobytes()
    env.close()


def test_sharding_independence():
    """Env with global id g behaves the same whether it lives in a batch starting at 0 or at g (per-rank offsets)."""
    from stratego_env_amd.vec_env import VecStrategoEnv
    seed = 99
    a = VecStrategoEnv('barrage', 96, seed=seed, env_id_offset=0, auto_reset=True)
    b = VecStrategoEnv('barrage', 32, seed=seed, env_id_offset=64, auto_reset=True)
    a.reset(); b.reset()
    a.sample_valid_actions(); b.sample_valid_actions()
    for t in range(200):
        a.rollout_step(); b.rollout_step()
    assert np.array_equal(a.obs[64:].cpu().numpy(), b.obs.cpu().numpy())
    assert np.array_equal(a.mask[64:].cpu().numpy(), b.mask.cpu().numpy())
    sa, pa = a.export_state(); sb, pb = b.export_state()
    assert np.array_equal(sa[64:].cpu().numpy(), sb.cpu().numpy())
    assert np.array_equal(a.env_info()[64:].cpu().numpy(), b.env_info().cpu().numpy())
    a.close(); b.close()


def test_full_size_rollout_properties_and_oracle_digests():
    """BASELINE config 2 size (65,536 Barrage games): size-independent properties on every env and the oracle's
    rolling digest on 256 sampled envs."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    N, T, seed = 65536, 48, 0x5712A7E60
    env = VecStrategoEnv('barrage', N, seed=seed, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    n_chk = 32
    ids = np.linspace(0, N - 1, n_chk).astype(np.int64)
    idx = torch.from_numpy(ids).to(env.device)
    digs = [orc.FNV_OFFSET] * n_chk
    fin = np.zeros(n_chk, dtype=np.int64)
    for t in range(T):
        env.rollout_step()
        m = env.mask.view(N, -1)
        assert int((m.sum(dim=1) == 0).sum()) == 0                       # always at least one action (no-op if stuck)
        assert int((m > 1).sum()) == 0
        o = env.obs.view(N, 100, 67)
        assert bool(torch.isfinite(o).all()) and float(o.abs().max()) <= 1.0
        assert bool((o[:, :, 0:12].sum(dim=2) <= 1).all())                # own-piece channels are one-hot
        assert int(env.invalid_action.sum()) == 0
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
        rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
        ei = env.ending_invalid[idx].cpu().numpy()
        for i in range(n_chk):
            # same byte stream as so_rollout(): mask, obs, reward +1, reward -1, {done, player, ending_invalid, error}
            tail = np.asarray([dn[i], pl[i], ei[i], 0], dtype=np.int32)
            digs[i] = orc.fnv1a(digs[i], mk[i].tobytes() + ob[i].tobytes() + rw[i].tobytes() + tail.tobytes())
            fin[i] += int(dn[i])
    # piece conservation on every one of the 65,536 exported states: pieces on the board + captured pieces = the variant's set;
    # what the opponent knows is the truth or UNKNOWN exactly where pieces stand; nothing overlaps
    st, _ = env.export_state()
    counts = torch.tensor(VARIANTS['barrage'].piece_counts, device=env.device)
    for pi, cap0 in ((0, 8), (1, 20)):
        on_board = torch.stack([(st[:, pi] == t).sum(dim=(1, 2)) for t in range(1, 13)], dim=1)
        captured = st[:, cap0:cap0 + 12].sum(dim=(2, 3))
        assert bool((on_board + captured == counts).all())
        known = st[:, 3 + pi]
        assert bool(((known != 0) == (st[:, pi] != 0)).all()) and bool(((known == st[:, pi]) | (known == 13)).all())
    assert not bool(((st[:, 0] != 0) & (st[:, 1] != 0)).any())
    cv = oracle_cvariant('barrage', setups=S.load_setup_table('barrage'))
    for i in range(n_chk):
        total, d, f = orc.rollout(cv, seed, int(ids[i]), 1, T, threads=1)
        assert int(d[0]) == digs[i], ('env', int(ids[i]))
        assert int(f[0]) == fin[i]
    env.close()


def test_standard_full_size_properties_and_oracle_digests():
    """BASELINE config 3 size (262,144 full-Stratego games): properties on every env + oracle digests of sampled games."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    N, T, seed = 262144, 24, 0x5712A7E60
    env = VecStrategoEnv('standard', N, seed=seed, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    n_chk = 12
    ids = np.linspace(0, N - 1, n_chk).astype(np.int64)
    idx = torch.from_numpy(ids).to(env.device)
    digs = [orc.FNV_OFFSET] * n_chk
    for t in range(T):
        env.rollout_step()
        m = env.mask.view(N, -1)
        assert int((m.sum(dim=1, dtype=torch.int32) == 0).sum()) == 0
        assert int(env.invalid_action.sum()) == 0
        if t % 8 == 0:
            o = env.obs.view(N, 100, 67)
            assert bool(torch.isfinite(o).all()) and float(o.abs().max()) <= 1.0
            assert bool((o[:, :, 0:12].sum(dim=2) <= 1).all())
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
        rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
        ei = env.ending_invalid[idx].cpu().numpy()
        for i in range(n_chk):
            tail = np.asarray([dn[i], pl[i], ei[i], 0], dtype=np.int32)
            digs[i] = orc.fnv1a(digs[i], mk[i].tobytes() + ob[i].tobytes() + rw[i].tobytes() + tail.tobytes())
    cv = oracle_cvariant('standard', setups=S.load_setup_table('standard'))
    for i in range(n_chk):
        total, d, f = orc.rollout(cv, seed, int(ids[i]), 1, T, threads=1)
        assert int(d[0]) == digs[i], ('env', int(ids[i]))
    env.close()


def test_placement_target_for_ring_sets():
    """sgx_set_placement_target: a search with a target goes on (inside its budget) until a candidate is within 3 % of it -- the
    sets of a ring should all be as fast as the first one -- and stops at the first candidate that is."""
    import ctypes as C
    from stratego_env_amd import _lib
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv('barrage', 4096, seed=5, auto_reset=True)
    env.reset()
    L = env._L
    assert L.sgx_set_placement_target(env._h, C.c_float(-1.0)) == -1           # SGX_EINVAL
    _lib.check(L.sgx_set_placement_target(env._h, C.c_float(1e9)), L)               # anything meets it: the plain allocation is kept
    assert len(env.tune_placement_once(trials=6)['obs']) == 1
    _lib.check(L.sgx_set_placement_target(env._h, C.c_float(1e-3)), L)              # nothing meets it: every candidate is tried
    assert len(env.tune_placement_once(trials=6)['obs']) == 6
    _lib.check(L.sgx_set_placement_target(env._h, C.c_float(0.0)), L)
    twin = VecStrategoEnv('barrage', 4096, seed=5, auto_reset=True)
    twin.reset()
    reps = env.alloc_output_ring(3, tune=True, trials=4)                            # target = what the env's own search kept
    assert reps[0] is None and all(1 <= len(r['obs']) <= 4 for r in reps[1:])
    env.rollout_steps(6, ring=True)
    twin.rollout_steps(6)
    import torch
    assert torch.equal(env.obs, twin.obs) and torch.equal(env.mask, twin.mask)
    # a set whose search ends above the target is searched once more over the wide budget and the faster of the two is kept: pretend
    # the env's own set was impossibly fast, so that every extra set takes that path
    env._outputs.trial_us[0] = 1e-3
    env._outputs.n_trials = 1
    reps = env.alloc_output_ring(3, tune=True, trials=3, max_extra_bytes=256 << 20, wide_extra_bytes=1 << 30)
    for r in reps[1:]:
        assert len(r['obs']) >= 3 and len(r['wide']['obs']) == 3 and r['wide']['used'] == (min(r['wide']['obs']) < min(r['obs'][:3]))
    env.rollout_steps(7, ring=True)
    twin.rollout_steps(7)
    assert torch.equal(env.obs, twin.obs) and torch.equal(env.mask, twin.mask) and torch.equal(env.reward, twin.reward)
    # the other way round: the env's own set looks 10 x slower than anything the extra sets find -> it is searched again against them,
    # replaced if that is faster, and the ring goes on with the right buffers and the current outputs
    env._outputs.trial_us[0] = 1e6
    env._outputs.n_trials = 1
    reps = env.alloc_output_ring(3, tune=True, trials=3, max_extra_bytes=256 << 20)
    assert reps[0] is not None and reps[0]['used'] and reps[0]['before_us'] == 1e6 and len(reps[0]['obs']) >= 1
    assert env._ring[0][0].data_ptr() == env.obs.data_ptr() == env._outputs.obs_dev
    assert torch.equal(env.obs, twin.obs) and torch.equal(env.mask, twin.mask)
    env.rollout_steps(5, ring=True)
    twin.rollout_steps(5)
    assert torch.equal(env.obs, twin.obs) and torch.equal(env.mask, twin.mask) and torch.equal(env.reward, twin.reward)
    env.close()
    twin.close()


def test_placement_search_as_a_constructor_argument(monkeypatch):
    """VecStrategoEnv(placement='search'), the default (SGX_PLACEMENT=plain switches it off): the first full reset() moves the outputs into
    buffers from the bounded search -- where there are placement classes (more than 300 MB of observations) -- and nothing else changes;
    small batches, compact outputs and record pools stay as they are; unknown values are refused."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    n = 12288                                                     # 329 MB of observations
    plain = VecStrategoEnv('barrage', n, seed=21, auto_reset=True, placement='plain')
    env = VecStrategoEnv('barrage', n, seed=21, auto_reset=True)                    # 'search' is the default
    assert env.placement_report is None and env._outputs is None
    plain.reset(); env.reset()
    assert env.placement_report and len(env.placement_report['obs']) >= 1 and env.obs.data_ptr() == env._outputs.obs_dev
    assert env.placement_peak_extra_bytes <= 8 << 30
    first = env.obs.data_ptr()
    for e in (plain, env):
        e.rollout_steps(40)
        e.reset(env_select=(torch.arange(n, device=e.device) % 3 == 0))      # (a partial reset never searches)
        e.rollout_steps(9)
        e.reset()
        e.rollout_steps(5)
    assert env.obs.data_ptr() == first and plain._outputs is None and plain.placement_report is None      # searched once; 'plain' never
    # the extra sets of a ring follow the env's own set: searched where that one was, plain tensors otherwise
    reps, plain_reps = env.alloc_output_ring(3), plain.alloc_output_ring(3)
    assert all(r and len(r['obs']) >= 1 for r in reps[1:]) and plain_reps == [None, None, None]
    for e in (plain, env):
        e.rollout_steps(11, ring=True)
    assert torch.equal(env.obs, plain.obs) and torch.equal(env.mask, plain.mask) and torch.equal(env.env_info(), plain.env_info())
    env.close(); plain.close()
    monkeypatch.setenv('SGX_PLACEMENT', 'plain')
    off = VecStrategoEnv('barrage', n, seed=1)
    off.reset()
    assert off.placement_report is None and off._outputs is None               # the environment variable switches the default off
    off.close()
    monkeypatch.setenv('SGX_PLACEMENT', 'search')
    small = VecStrategoEnv('barrage', 512, seed=1)
    small.reset()
    assert small.placement_report is None and small._outputs is None           # nothing to search for below 300 MB
    small.close()
    compact = VecStrategoEnv('barrage', n, seed=1, compact_outputs=True)
    compact.reset()
    assert compact.placement_report is None
    compact.close()
    pool = VecStrategoEnv('barrage', 4096, seed=1, outputs=False)
    pool.close()
    monkeypatch.setenv('SGX_PLACEMENT', 'fastest')
    with pytest.raises(ValueError):
        VecStrategoEnv('barrage', 8, seed=1)
    monkeypatch.delenv('SGX_PLACEMENT')
    with pytest.raises(ValueError):
        VecStrategoEnv('barrage', 8, seed=1, placement='auto')


def test_tune_placement_keeps_outputs():
    """Placement trials only swap which allocation the outputs live in."""
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv('barrage', 4096, seed=5, auto_reset=True)
    env.reset()
    obs0, mask0 = env.obs.clone(), env.mask.clone()
    rep = env.tune_placement(trials=3)
    assert len(rep['obs']) == 3 and env.placement_peak_extra_bytes <= 8 << 30
    assert env.obs.data_ptr() == env._outputs.obs_dev and env.mask.data_ptr() == env._outputs.mask_dev     # library-owned buffers
    assert np.array_equal(env.obs.cpu().numpy(), obs0.cpu().numpy()) and np.array_equal(env.mask.cpu().numpy(), mask0.cpu().numpy())
    env.sample_valid_actions()
    env.rollout_step()
    assert int(env.invalid_action.sum()) == 0
    # tensors handed out keep the library-owned memory alive: after close() / del / another tune_placement() they still read what
    # they held (the buffers are freed with the last tensor that views them)
    import gc
    import torch
    obs_kept, mask_kept = env.rollout_steps(3)[:2]
    want_obs, want_mask = obs_kept.clone(), mask_kept.clone()
    env.tune_placement(trials=2)                                       # the env moves on to other buffers
    assert env.obs.data_ptr() != obs_kept.data_ptr()
    env.close()
    del env
    gc.collect()
    junk = [torch.full((64 << 20,), 7.0, device='cuda') for _ in range(8)]        # anything freed would be reused and overwritten here
    torch.cuda.synchronize()
    assert torch.equal(obs_kept, want_obs) and torch.equal(mask_kept, want_mask)
    del junk, obs_kept, mask_kept
    gc.collect()
    # the wide second pass (taken when the first budget held no candidate >= 14 % below its slowest one, streaming launches only):
    # whichever pass is kept, the env goes on with valid buffers that hold the current outputs
    env = VecStrategoEnv('barrage', 16384, seed=6, auto_reset=True)         # 439 MB of observations: past the Infinity Cache
    twin = VecStrategoEnv('barrage', 16384, seed=6, auto_reset=True)
    env.reset()
    twin.reset()
    rep = env.tune_placement(trials=2, max_extra_bytes=1 << 30, wide_extra_bytes=3 << 30)
    if 'wide' in rep:
        assert len(rep['wide']['obs']) == 2 and rep['wide']['used'] == (min(rep['wide']['obs']) < min(rep['obs']))
    assert env.obs.data_ptr() == env._outputs.obs_dev and torch.equal(env.obs, twin.obs) and torch.equal(env.mask, twin.mask)
    env.rollout_steps(5)
    twin.rollout_steps(5)
    assert torch.equal(env.obs, twin.obs) and torch.equal(env.mask, twin.mask)
    env.close()
    twin.close()
    env = VecStrategoEnv('tiny', 1024, seed=5, auto_reset=True, full_obs=True)       # BOTH mode: the full observation is placed too
    env.reset()
    fobs0 = env.fobs.clone()
    rep = env.tune_placement(trials=2)
    assert sorted(rep) == ['fobs', 'obs'] and len(rep['fobs']) == 2 and np.array_equal(env.fobs.cpu().numpy(), fobs0.cpu().numpy())
    rep = env.tune_placement(trials=4, max_extra_bytes=0)                   # no budget: plain first allocation, nothing timed
    assert rep['obs'] == [] and np.array_equal(env.fobs.cpu().numpy(), fobs0.cpu().numpy())
    env.sample_valid_actions()
    env.rollout_step()
    assert int(env.invalid_action.sum()) == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['micro', 'barrage'])
def test_step_n_equals_n_single_steps(name):
    """sgx_step_n (k rollout steps in one library call) leaves exactly the state and outputs of k sgx_step calls."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    from stratego_env_amd import _lib
    import ctypes as C
    k, n = 37, 512
    envs = [VecStrategoEnv(name, n, seed=0xABCD, auto_reset=True) for _ in range(2)]
    for e in envs:
        e.reset()
        e.sample_valid_actions()
    for _ in range(k):
        envs[0].rollout_step()
    envs[1].rollout_steps(k)
    torch.cuda.synchronize()
    for attr in ('obs', 'mask', 'reward', 'done', 'player', 'invalid_action', 'ending_invalid', 'next_actions'):
        assert torch.equal(getattr(envs[0], attr), getattr(envs[1], attr)), attr
    s0, p0 = envs[0].export_state()
    s1, p1 = envs[1].export_state()
    assert torch.equal(s0, s1) and torch.equal(p0, p1)
    assert torch.equal(envs[0].env_info(), envs[1].env_info())
    # the call insists on the self-feeding configuration
    io = envs[1]._fill_io(envs[1].next_actions, False, True, True, 0)
    assert envs[1]._L.sgx_step_n(envs[1]._h, C.byref(io), 3, None) == -1
    assert b'next_actions_dev' in envs[1]._L.sgx_last_error()
    for e in envs:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,N', [('micro', 65536 + 17), ('tiny', 40003), ('fives', 30001)])
def test_toy_full_size_shared_waves_vs_oracle_digests(name, N):
    """BASELINE config 4 size (65,536 Micro games) plus a ragged tail: toy boards share a wave (4 or 2 games per wave), so the
    sampled envs include the first and last games of waves and of the partly filled last workgroup.  Micro games last ~11
    moves: 64 steps are ~6 auto-resets per env."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    T, seed = 64, 0xD1CE
    v = VARIANTS[name]
    env = VecStrategoEnv(name, N, seed=seed, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    ids = np.unique(np.concatenate([np.arange(0, 9), np.arange(28, 36), np.arange(N - 19, N),
                                    np.linspace(0, N - 1, 24).astype(np.int64)]))
    idx = torch.from_numpy(ids).to(env.device)
    digs = [orc.FNV_OFFSET] * len(ids)
    cells = v.rows * v.columns
    for t in range(T):
        env.rollout_step()
        m = env.mask.view(N, -1)
        assert int((m.sum(dim=1, dtype=torch.int32) == 0).sum()) == 0 and int((m > 1).sum()) == 0
        o = env.obs.view(N, cells, 67)
        assert bool(torch.isfinite(o).all()) and float(o.abs().max()) <= 1.0
        assert int(env.invalid_action.sum()) == 0
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
        rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
        ei = env.ending_invalid[idx].cpu().numpy()
        for i in range(len(ids)):
            tail = np.asarray([dn[i], pl[i], ei[i], 0], dtype=np.int32)
            digs[i] = orc.fnv1a(digs[i], mk[i].tobytes() + ob[i].tobytes() + rw[i].tobytes() + tail.tobytes())
    cv = oracle_cvariant(name)
    for i in range(len(ids)):
        total, d, f = orc.rollout(cv, seed, int(ids[i]), 1, T, threads=1)
        assert int(d[0]) == digs[i], (name, 'env', int(ids[i]))
    assert int(env.env_info()[:, 1].sum()) > N          # games were finished and restarted everywhere
    env.close()


@pytest.mark.gpu
def test_calls_run_on_the_callers_stream():
    """Every entry point enqueues on the stream it is given (torch's current stream): a rollout issued inside a side-stream
    context, with the default stream kept busy, equals the same rollout on the default stream."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    n, k = 2048, 25
    a = VecStrategoEnv('barrage', n, seed=99, auto_reset=True)
    b = VecStrategoEnv('barrage', n, seed=99, auto_reset=True)
    a.reset(); a.sample_valid_actions()
    for _ in range(k):
        a.rollout_step()
    side = torch.cuda.Stream()
    busy = torch.empty(1 << 26, device='cuda')
    with torch.cuda.stream(side):
        b.reset(); b.sample_valid_actions()
        for _ in range(k):
            busy.fill_(1.0)                        # default-stream work that must not order b's launches
            b.rollout_step()
        st_b, pl_b = b.export_state()
    side.synchronize()
    torch.cuda.synchronize()
    st_a, pl_a = a.export_state()
    assert torch.equal(st_a, st_b) and torch.equal(pl_a, pl_b)
    assert torch.equal(a.obs, b.obs) and torch.equal(a.mask, b.mask) and torch.equal(a.next_actions, b.next_actions)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['barrage', 'standard', 'micro', 'fives'])
def test_import_sanitises_unreachable_states(name):
    """sgx_import_state accepts any int64 tensor: values outside a layer's legal range become 0, captured counts are
    clamped to 8, at most two recent-move cells per player and at most max_events (layer, cell) pairs with captures survive
    (documented limits of the packed record) -- and stepping such states afterwards stays in bounds (no invalid memory access, outputs finite)."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[name]
    R, C = v.rows, v.columns
    n = 64
    rng = np.random.RandomState(7)
    st = rng.randint(-5, 40, size=(n, 34, R, C)).astype(np.int64)
    st[:, 5] = 0
    st[:, 5, 0, 0] = rng.randint(0, 10, size=n)           # turn count
    st[:, 5, 1, 0] = v.max_turns
    env = VecStrategoEnv(name, n, seed=1, auto_reset=True)
    players = np.where(rng.rand(n) < 0.5, 1, -1).astype(np.int8)
    env.import_state(torch.from_numpy(st), torch.from_numpy(players))
    out, pl = env.export_state()
    out = out.cpu().numpy()
    assert np.array_equal(pl.cpu().numpy(), players)
    max_events = 2 * sum(v.piece_counts)
    for e in range(n):
        for layer, hi in ((0, 12), (1, 12), (3, 13), (4, 13)):
            want = np.where((st[e, layer] >= 0) & (st[e, layer] <= hi), st[e, layer], 0)
            assert np.array_equal(out[e, layer], want), (name, e, layer)
        for layer in (32, 33):
            assert np.array_equal(out[e, layer], (st[e, layer] == 1).astype(np.int64))
        for layer in (6, 7):                                # first two legal non-zero codes in cell order
            codes = np.where((st[e, layer] >= -3) & (st[e, layer] <= 1), st[e, layer], 0).reshape(-1)
            keep = np.flatnonzero(codes)[:2]
            want = np.zeros(R * C, dtype=np.int64)
            want[keep] = codes[keep]
            assert np.array_equal(out[e, layer].reshape(-1), want), (name, e, layer)
        caps = np.clip(st[e, 8:32], 0, 8).reshape(-1)       # at most 8 per (layer, cell); the first max_events non-zero pairs survive
        want = np.zeros_like(caps)
        keep = np.flatnonzero(caps)[:max_events]
        want[keep] = caps[keep]
        assert np.array_equal(out[e, 8:32].reshape(-1), want), (name, e, 'captured')
        assert np.array_equal(out[e, 2], v.obstacle_map().astype(np.int64))
    env.sample_valid_actions()
    for _ in range(8):
        env.rollout_step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(env.obs).all()) and int((env.mask > 1).sum()) == 0
    env.close()


@pytest.mark.gpu
def test_long_horizon_rollout_with_max_turn_endings():
    """2,500 batched steps of 4,096 Barrage games: several auto-resets per env, games that run into max_turns = 1000
    (ENDING_INVALID, impl:1040-1043) inside the batch, capture-event lists that fill up -- oracle digests of sampled envs."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    N, T, seed = 4096, 2500, 0xFEED5
    env = VecStrategoEnv('barrage', N, seed=seed, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    ids = np.asarray([0, 7, 1000, 2047, 4095], dtype=np.int64)
    idx = torch.from_numpy(ids).to(env.device)
    digs = [orc.FNV_OFFSET] * len(ids)
    invalid_endings = torch.zeros((), dtype=torch.int64, device=env.device)
    for t in range(T):
        env.rollout_step()
        invalid_endings += env.ending_invalid.sum()
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
        rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
        ei = env.ending_invalid[idx].cpu().numpy()
        for i in range(len(ids)):
            tail = np.asarray([dn[i], pl[i], ei[i], 0], dtype=np.int32)
            digs[i] = orc.fnv1a(digs[i], mk[i].tobytes() + ob[i].tobytes() + rw[i].tobytes() + tail.tobytes())
    assert int(env.invalid_action.sum()) == 0
    assert int(invalid_endings) > 0                       # max-turn endings did occur
    cv = oracle_cvariant('barrage', setups=S.load_setup_table('barrage'))
    for i in range(len(ids)):
        total, d, f = orc.rollout(cv, seed, int(ids[i]), 1, T, threads=1)
        assert int(d[0]) == digs[i], ('env', int(ids[i]))
    info = env.env_info().cpu().numpy()
    assert info[:, 1].min() >= 2                          # every env finished at least two games
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['tiny', 'barrage'])
def test_manual_partial_resets_match_oracle(name):
    """The non-auto-reset loop: finished envs stay finished (stepping them is an invalid action, state unchanged) until the
    caller resets exactly those through `env_select`; a sampled reset starts the env's next game number, other envs are
    untouched."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    n, T = 48, 260 if name == 'tiny' else 1300
    seed, g0 = 0x51DE + len(name), 77
    env = VecStrategoEnv(name, n, seed=seed, env_id_offset=g0, auto_reset=False)
    cv, oenvs = _oracle_batch(name, seed, g0, n)
    obs, mask, player = env.reset()
    cur = [oe._obs(1) for oe in oenvs]
    finished = np.zeros(n, dtype=bool)
    resets = 0
    for t in range(T):
        acts = np.zeros(n, dtype=np.int32)
        for e, oe in enumerate(oenvs):
            acts[e] = orc.sample_action(cur[e][MASK].astype(np.uint8), seed, g0 + e, oe.game_no, int(oe.state[5, 0, 0]))
        env.step(torch.from_numpy(acts))
        done_h, inv_h = env.done.cpu().numpy(), env.invalid_action.cpu().numpy()
        obs_h, mask_h, player_h = env.obs.cpu().numpy(), env.mask.cpu().numpy(), env.player.cpu().numpy()
        for e, oe in enumerate(oenvs):
            if finished[e]:
                # a finished game offers only the no-op; whatever the reference does with it (the oracle is pinned to the
                # reference on post-terminal steps too), the batched env does the same and the state stays where it was
                assert int(mask_h[e].sum()) == 1 and mask_h[e][0, 0, -1] == 1
                before = oe.state.copy()
                try:
                    oe.step({oe.player: int(acts[e])})
                    assert inv_h[e] == 0
                except ValueError:
                    assert inv_h[e] == 1
                assert np.array_equal(oe.state, before)
                continue
            o, rew, done, info = oe.step({oe.player: int(acts[e])})
            assert inv_h[e] == 0 and bool(done_h[e]) == done['__all__']
            if done['__all__']:
                finished[e] = True
                cur[e] = {MASK: o[oe.player][MASK] * 0}
                cur[e][MASK][0, 0, -1] = 1
            else:
                p = oe.player
                assert player_h[e] == p and np.array_equal(o[p][MASK], mask_h[e]) and o[p][POBS].tobytes() == obs_h[e].tobytes()
                cur[e] = o[p]
        if t % 40 == 39 and finished.any():
            before, _ = env.export_state()
            sel = torch.from_numpy(finished.astype(np.uint8))
            obs, mask, player = env.reset(env_select=sel)
            after, _ = env.export_state()
            keep = torch.from_numpy(~finished).to(env.device)
            assert torch.equal(before[keep], after[keep])                       # unselected envs untouched
            obs_h, mask_h = obs.cpu().numpy(), mask.cpu().numpy()
            for e in np.flatnonzero(finished):
                oe = oenvs[e]
                oe.game_no += 1
                o = oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, oe.game_no))
                assert np.array_equal(after[e].cpu().numpy(), oe.state)
                assert np.array_equal(o[1][MASK], mask_h[e]) and o[1][POBS].tobytes() == obs_h[e].tobytes()
                cur[e] = o[1]
                resets += 1
            finished[:] = False
    assert resets > 0
    env.close()


@pytest.mark.gpu
def test_rollout_steps_draws_first_actions_itself():
    """rollout_steps() right after reset() (no explicit sample_valid_actions) plays valid moves from the first step on and
    equals the explicit sequence."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv('micro', 300, seed=4, auto_reset=True)
    b = VecStrategoEnv('micro', 300, seed=4, auto_reset=True)
    a.reset(); a.rollout_steps(30)
    b.reset(); b.sample_valid_actions()
    for _ in range(30):
        b.rollout_step()
    assert int(a.invalid_action.sum()) == 0
    assert torch.equal(a.obs, b.obs) and torch.equal(a.next_actions, b.next_actions) and torch.equal(a.env_info(), b.env_info())
    a.reset(); a.rollout_step()                     # a reset invalidates the pending draw
    assert int(a.invalid_action.sum()) == 0
    a.close(); b.close()


@pytest.mark.gpu
def test_cabi_error_returns_on_device():
    """Negative return codes + sgx_last_error text instead of exceptions or crashes; nothing is created on failure."""
    import ctypes as C
    from stratego_env_amd import _lib
    L = _lib.load()
    cfg = _lib.make_config(VARIANTS['barrage'])
    h = C.c_void_p()
    assert L.sgx_create(C.byref(cfg), 0, 0, 1, 0, C.byref(h)) == -1 and not h.value and b'n_envs' in L.sgx_last_error()
    assert L.sgx_create(C.byref(cfg), 16, 99, 1, 0, C.byref(h)) == -1 and not h.value and b'device' in L.sgx_last_error()
    bad = _lib.make_config(VARIANTS['barrage'])
    bad.rows, bad.cols = 7, 7                                   # a size no reference variant uses: not in the MAIN library
    bad.usable_rows = 2                                         # (build_geometry gives it a library of its own)
    assert L.sgx_supports_geometry(7, 7) == 0 and L.sgx_supports_geometry(10, 10) == 1
    assert L.sgx_create(C.byref(bad), 16, 0, 1, 0, C.byref(h)) == -1 and not h.value and b'not compiled into this library' in L.sgx_last_error()
    assert L.sgx_create(C.byref(cfg), 16, 0, 1, 0, None) == -1
    assert L.sgx_create(C.byref(cfg), 16, 0, 1, 0, C.byref(h)) == 0 and h.value
    assert L.sgx_step(h, None, None) == -1
    io = _lib.SgxStepIO()
    assert L.sgx_step(h, C.byref(io), None) == -1 and b'actions_dev' in L.sgx_last_error()
    assert L.sgx_reset(h, None, C.c_void_p(1), None, None) == -1 and b'both piece maps' in L.sgx_last_error()
    assert L.sgx_sample_valid(h, None, None, None) == -1
    assert L.sgx_export_state(h, None, None, None) == -1 and L.sgx_import_state(h, None, None, None) == -1
    tab = (C.c_uint8 * 40)(*([13] + [0] * 39))                   # piece code 13 is not a piece
    assert L.sgx_set_setup_table(h, tab, 1) == -1 and b'piece code' in L.sgx_last_error()
    assert L.sgx_num_envs(h) == 16 and L.sgx_spatial_channels(h) == 37 and L.sgx_num_spatial_actions(h) == 3700
    assert L.sgx_action_size_1d(h) == 2001
    assert L.sgx_destroy(h) == 0 and L.sgx_destroy(None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('name,n', [('micro', 4096), ('barrage', 1024), ('fives', 1500)])
def test_rollout_chains_equal_a_single_chain(name, n):
    """sgx_rollout: the batch split into 2 / 3 / 4 ranges of games playing on streams of their own leaves exactly the state and the
    outputs of sgx_step_n (ragged batch sizes included: the last range takes the remainder)."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    ref = VecStrategoEnv(name, n, seed=77, auto_reset=True)
    ref.reset()
    ref.rollout_steps(41)
    want_state, want_player = ref.export_state()
    for chains in (2, 3, 4, 0, 'auto'):
        env = VecStrategoEnv(name, n, seed=77, auto_reset=True)
        env.reset()
        env.rollout_steps(20, chains=chains)
        env.rollout_steps(21, chains=chains)
        torch.cuda.synchronize()
        st, pl = env.export_state()
        assert torch.equal(st, want_state) and torch.equal(pl, want_player), (name, chains)
        for a, b in ((env.obs, ref.obs), (env.mask, ref.mask), (env.reward, ref.reward), (env.done, ref.done), (env.next_actions, ref.next_actions)):
            assert torch.equal(a, b), (name, chains)
        assert int(env.invalid_action.sum()) == 0
        env.close()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,n,full', [('barrage', 1024, False), ('micro', 4096, False), ('fives', 1500, False), ('barrage', 512, True)])
def test_step_ring_equals_step_n(name, n, full):
    """sgx_step_ring: the same rollout writing a ring of output sets round-robin leaves exactly the state of sgx_step_n; the set the
    last step wrote holds sgx_step_n's outputs, and the sets before it hold the outputs of the steps before."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    per_step = []
    ref = VecStrategoEnv(name, n, seed=91, auto_reset=True, full_obs=full)
    ref.reset()
    for _ in range(11):
        ref.rollout_steps(1)
        per_step.append((ref.obs.clone(), ref.mask.clone(), ref.fobs.clone() if full else None))
    want_state, want_player = ref.export_state()
    for n_sets in (1, 2, 3, 4):
        env = VecStrategoEnv(name, n, seed=91, auto_reset=True, full_obs=full)
        env.reset()
        env.alloc_output_ring(n_sets)
        env.rollout_steps(4, ring=True)
        env.rollout_steps(7, ring=True)
        torch.cuda.synchronize()
        st, pl = env.export_state()
        assert torch.equal(st, want_state) and torch.equal(pl, want_player), (name, n_sets)
        for a, b in ((env.obs, ref.obs), (env.mask, ref.mask), (env.reward, ref.reward), (env.done, ref.done), (env.next_actions, ref.next_actions)):
            assert torch.equal(a, b), (name, n_sets)
        if full:
            assert torch.equal(env.fobs, ref.fobs)
        # set (1 + i) mod n_sets was written by step i: the last n_sets steps are all still there
        for back in range(min(n_sets, 11)):
            i = 10 - back
            obs, mask, fobs = env._ring[(1 + i) % n_sets]
            assert torch.equal(obs, per_step[i][0]) and torch.equal(mask, per_step[i][1]), (name, n_sets, back)
            if full:
                assert torch.equal(fobs, per_step[i][2])
        assert int(env.invalid_action.sum()) == 0
        env.close()
    ref.close()


@pytest.mark.gpu
def test_record_bytes_match_the_bench_arithmetic():
    """bench.py prices its roofline on B_min, which contains the packed record twice: its own arithmetic must be the library's."""
    import bench
    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.vec_env import VecStrategoEnv
    for name, v in VARIANTS.items():
        env = VecStrategoEnv(name, 8, seed=1)
        assert env.record_bytes == bench.record_bytes(v) and env.record_bytes % 128 == 0, name
        assert len(env.build_id) == 16
        env.close()


@pytest.mark.gpu
def test_entry_points_leave_the_callers_device_alone():
    """Every entry point runs on its handle's device and restores the caller's (DeviceGuard).  With one GPU the current device cannot
    differ from the handle's, so this checks the calls through a second, never-selected ordinal only where the box has one; on any
    box: the HIP runtime's current device is what it was before each call."""
    import ctypes
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    hip = ctypes.CDLL('libamdhip64.so')
    cur = ctypes.c_int(-1)

    def current():
        assert hip.hipGetDevice(ctypes.byref(cur)) == 0
        return cur.value
    ndev = torch.cuda.device_count()
    home = 0
    target = 1 if ndev > 1 else 0
    torch.cuda.set_device(home)
    assert current() == home
    env = VecStrategoEnv('barrage', 64, device=target, seed=3, auto_reset=True)
    assert current() == home
    env.reset(); assert current() == home
    env.rollout_steps(5); assert current() == home
    env.rollout_steps(4, chains=2); assert current() == home
    st, pl = env.export_state(); assert current() == home
    env.import_state(st, pl); assert current() == home
    env.tune_placement(trials=2, max_extra_bytes=1 << 30); assert current() == home
    keep = env.obs
    env.close(); assert current() == home
    del keep, env
    import gc
    gc.collect()
    assert current() == home
    if ndev > 1:
        # results on the other device equal results on the home device
        a = VecStrategoEnv('barrage', 64, device=0, seed=3, auto_reset=True); a.reset(); a.rollout_steps(9)
        b = VecStrategoEnv('barrage', 64, device=1, seed=3, auto_reset=True); b.reset(); b.rollout_steps(9)
        assert torch.equal(a.obs.cpu(), b.obs.cpu()) and torch.equal(a.mask.cpu(), b.mask.cpu())
        a.close(); b.close()
