"""sgx_choose_actions -- the device-side counterpart of the reference's chooser (examples/basic_game_loop.py:6-31 with softmax of
examples/util.py:4-47): masked softmax over caller logits, one sample per game with the env's counter RNG.

Pinned by: (1) equal logits reproduce the fused sampler / sgx_sample_valid action for action (the k-th valid one with the same draw),
on every board size, byte and bit masks; (2) a NumPy restatement of the kernel's fixed-point inverse CDF (float64 exp2 agrees with
v_exp_f32 except within an ulp of a weight boundary, where either neighbour is accepted); (3) a masked action is never chosen and
the empirical distribution over many games with the same logits follows softmax (chi-square); (4) temperature 0 is argmax, NaN / -inf
logits are never chosen, an empty mask gives -1; (5) a whole rollout driven by the chooser replays on the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BITS = 23


def _rng_draws(env, ids):
    """r32 of the counter RNG for the current (game, turn) of the envs `ids` (the oracle's restatement of sgx_rng)."""
    from oracle import oracle as orc
    info = env.env_info().cpu().numpy()
    return np.asarray([orc.rng(env.seed, env.env_id_offset + int(e), int(info[e, 1]), 1, int(info[e, 0])) >> 32 for e in ids], dtype=np.uint64)


def _expected(logits, mask, r32, temperature, slack=None):
    """The kernel's rule in NumPy for one game -> set of acceptable actions (one, unless a weight sits within rounding of an integer)."""
    valid = (mask != 0) & ~np.isnan(logits)
    l = np.where(valid, logits.astype(np.float32), -np.inf).astype(np.float32)
    mx = l.max()
    if mx == -np.inf:
        return {-1}
    scale = np.float32(np.inf) if temperature == 0 else np.float32(1.4426950408889634) / np.float32(temperature)
    with np.errstate(invalid='ignore', over='ignore'):
        x = np.float32(l - mx) * scale + np.float32(BITS)          # (the kernel uses an fma; one rounding less: within the tolerance below)
        w_exact = np.exp2(x.astype(np.float64))
    w_exact = np.where(l == mx, float(1 << BITS), np.where(np.isnan(w_exact), 0.0, w_exact))
    outs = set()
    for delta in (0.0, -1.0, 1.0):          # every weight nudged by an ulp-sized amount: the set of answers rounding could give
        w = np.floor(np.where(l == mx, w_exact, np.maximum(w_exact * (1.0 + delta * 2.0 ** -21), 0.0))).astype(np.uint64)
        total = int(w.sum())
        target = (int(r32) * total) >> 32
        cum = np.cumsum(w.astype(object))
        outs.add(int(np.argmax(cum > target)))
    return outs


@pytest.mark.parametrize('name,n', [('barrage', 257), ('standard', 96), ('octa_barrage', 130), ('medium', 300), ('fives', 301), ('tiny', 515),
                                    ('micro', 1001), ('standard2', 40)])
def test_equal_logits_reproduce_the_uniform_sampler_exactly(name, n):
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    for compact in (False, True):
        env = VecStrategoEnv(name, n, seed=4242, env_id_offset=77, auto_reset=True, compact_outputs=compact)
        env.reset()
        logits = torch.full((n, env.R, env.Cc, env.K), 0.25, dtype=torch.float32, device=env.device)
        scratch = torch.empty((n,), dtype=torch.int32, device=env.device)
        for t in range(40):
            got = env.choose_actions(logits, temperature=0.7 + 0.1 * (t % 5), out=scratch)
            want = env.sample_valid_actions()                       # sgx_sample_valid: the k-th valid action with the same draw
            assert torch.equal(got, want), (name, compact, t)
            env.step(got, want_next_actions=True)
            assert int(env.invalid_action.sum()) == 0
            assert torch.equal(env.choose_actions(logits, out=scratch), env.next_actions)      # ... and the fused sampler's
        env.close()


@pytest.mark.parametrize('name,n', [('barrage', 200), ('fives', 150), ('micro', 300), ('standard2', 24)])
def test_choice_follows_the_fixed_point_rule(name, n):
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv(name, n, seed=99, env_id_offset=3, auto_reset=True)
    env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(5)
    for t in range(12):
        logits = torch.randn((n, env.R * env.Cc * env.K), device=env.device, generator=g) * (1.0 + t)
        if t % 3 == 1:                                              # special values: NaN and -inf are never chosen, +inf wins
            logits[:, ::7] = float('nan')
            logits[:, 3::11] = float('-inf')
            logits[::5, 5::13] = float('inf')
        temp = [1.0, 0.5, 2.0, 0.0][t % 4]
        got = env.choose_actions(logits, temperature=temp).cpu().numpy()
        lg, mk = logits.cpu().numpy(), env.mask.reshape(n, -1).cpu().numpy()
        r32 = _rng_draws(env, range(n))
        for e in range(n):
            acc = _expected(lg[e], mk[e], r32[e], temp)
            assert int(got[e]) in acc, (name, t, e, int(got[e]), acc)
            if got[e] >= 0:
                assert mk[e, got[e]] != 0 and not np.isnan(lg[e, got[e]]) and lg[e, got[e]] != -np.inf
                if temp == 0.0:
                    assert lg[e, got[e]] == np.nanmax(np.where(mk[e] != 0, lg[e], -np.inf))
        env.step(torch.from_numpy(np.maximum(got, 0)).to(env.device))
    # an empty mask, and a mask whose only valid logits are -inf / NaN: -1
    empty = torch.zeros_like(env.mask)
    assert bool((env.choose_actions(torch.zeros((n, env.R * env.Cc * env.K), device=env.device), mask=empty) == -1).all())
    dead = torch.full((n, env.R * env.Cc * env.K), float('-inf'), device=env.device)
    dead[:, ::2] = float('nan')
    assert bool((env.choose_actions(dead) == -1).all())
    env.close()


def test_distribution_is_the_softmax_over_valid_actions():
    """65,536 Barrage games in the SAME position (fresh games from one setup pair) with the same logits: the draws differ only by the
    counter RNG's key (the global env id); the empirical distribution must follow softmax over the valid actions (chi-square)."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    n = 65536
    env = VecStrategoEnv('barrage', n, seed=7, auto_reset=False, human_inits=False)
    m1 = np.zeros((10, 10), dtype=np.int8)
    m1[3, 0], m1[3, 4], m1[3, 5], m1[2, 2], m1[0, 0], m1[1, 1], m1[3, 9], m1[2, 8] = 2, 2, 3, 9, 11, 12, 1, 10
    maps = torch.from_numpy(np.broadcast_to(m1, (n, 10, 10)).copy())
    env.reset(maps, maps)
    mask0 = env.mask[0].reshape(-1).cpu().numpy()
    valid = np.flatnonzero(mask0)
    assert len(valid) >= 10 and bool((env.mask.reshape(n, -1) == env.mask[0].reshape(1, -1)).all())
    rs = np.random.RandomState(3)
    row = rs.randn(mask0.size).astype(np.float32) * 1.5
    logits = torch.from_numpy(np.broadcast_to(row, (n, mask0.size)).copy()).to(env.device)
    for temp in (1.0, 0.6):
        got = env.choose_actions(logits, temperature=temp).cpu().numpy()
        assert np.isin(got, valid).all()                            # never a masked action
        z = row[valid].astype(np.float64) / temp
        p = np.exp(z - z.max())
        p /= p.sum()
        counts = np.asarray([(got == a).sum() for a in valid], dtype=np.float64)
        keep = n * p >= 5
        chi2 = (((counts - n * p) ** 2) / (n * p))[keep].sum()
        dof = int(keep.sum()) - 1
        assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (temp, chi2, dof)
    env.close()


def test_rollout_driven_by_the_chooser_replays_on_the_oracle():
    """basic_game_loop for a batch, chooser on the device: the logged actions of sampled envs replay on the CPU oracle step for step
    (auto-reset included) and the last step's mask / observation / rewards match bit for bit."""
    import torch
    from tests.helpers import oracle_cvariant
    from oracle import oracle as orc
    from stratego_env_amd import setups as S
    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.vec_env import VecStrategoEnv
    name, n, T, seed = 'barrage', 512, 150, 0xC0FFEE
    v = VARIANTS[name]
    env = VecStrategoEnv(name, n, seed=seed, auto_reset=True)
    env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(11)
    readout = torch.randn(67, env.R * env.Cc * env.K, device=env.device, generator=g)
    ids = [0, 1, 100, 511]
    acts, dones = np.zeros((T, len(ids)), dtype=np.int64), np.zeros((T, len(ids)), dtype=np.uint8)
    for t in range(T):
        logits = env.obs.mean(dim=(1, 2)) @ readout
        a = env.choose_actions(logits, temperature=0.8)
        acts[t] = a[ids].cpu().numpy()
        env.step(a)
        dones[t] = env.done[ids].cpu().numpy()
    assert int(env.invalid_action.sum()) == 0
    cv = oracle_cvariant(name, setups=S.load_setup_table('barrage'))
    for c, e in enumerate(ids):
        oe = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts)
        game = 0
        obs = oe.reset(initial_state_override=orc.reset_state(cv, seed, e, game))
        for t in range(T):
            o, r, d, info = oe.step({oe.player: int(acts[t, c])})
            assert bool(d['__all__']) == bool(dones[t, c]), (e, t)
            if d['__all__']:
                game += 1
                first = oe.reset(initial_state_override=orc.reset_state(cv, seed, e, game))
                last_obs = first[1]
            else:
                last_obs = o[oe.player]
        assert np.array_equal(last_obs[oe.MASK].astype(np.uint8), env.mask[e].cpu().numpy())
        assert last_obs[oe.POBS].tobytes() == env.obs[e].cpu().numpy().tobytes()
    env.close()


def test_argument_checks():
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv('tiny', 8, seed=1)
    env.reset()
    lg = torch.zeros((8, env.R * env.Cc * env.K), device=env.device)
    with pytest.raises(ValueError):
        env.choose_actions(lg[:, :-1])
    for bad in (-1.0, float('nan'), float('inf')):
        with pytest.raises(_lib.SgxError):
            env.choose_actions(lg, temperature=bad)
    # a misaligned logits tensor takes the element-wise kernel and gives the same answer
    big = torch.zeros((8 * env.R * env.Cc * env.K + 1,), device=env.device)
    a = env.choose_actions(lg, out=torch.empty(8, dtype=torch.int32, device=env.device))
    b = env.choose_actions(big[1:].view(8, -1), out=torch.empty(8, dtype=torch.int32, device=env.device))
    assert torch.equal(a, b)
    env.close()
