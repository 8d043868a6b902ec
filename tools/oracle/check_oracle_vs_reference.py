"""Differential check: oracle/stratego_oracle.c vs the imported reference (BUILD CONTAINER ONLY).

Plays full games of every variant through both StrategoMultiAgentEnv (reference, stub-imported) and
oracle.OracleEnv with identical setups and actions -- valid random actions plus injected garbage
actions (every flat index class incl. the spatial no-op, out-of-range, negative) -- and compares, at
every step, error behaviour, obs dict keys, mask, partial-observation BYTES, rewards, dones, infos and
the full int64 state.  Also checks the index algebra exhaustively and the pure functions on the
visited states.  Exit code 0 = oracle pinned to the reference.

    python -m tools.oracle.check_oracle_vs_reference [--games N]
"""
import argparse
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.oracle.ref_stubs import import_reference  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def own_side_maps_from_state(state):
    """Invert _create_initial_state (impl:211-249): absolute state -> the two own-side piece maps."""
    return state[0].copy(), state[1][::-1, ::-1].copy()


def check_index_algebra(ref, R, C):
    pe = ref.penv.StrategoProceduralEnv(R, C)
    ru = orc.OracleRules(R, C)
    assert int(pe.action_size) == ru.action_size
    assert tuple(int(x) for x in pe.spatial_action_size) == ru.spatial_action_size
    K = ru.K
    for r in range(R):
        for c in range(C):
            for ch in range(K):
                a = tuple(int(x) for x in pe.get_action_positions_from_spatial_index((r, c, ch)))
                b = ru.get_action_positions_from_spatial_index((r, c, ch))
                assert a == b, (r, c, ch, a, b)
                i1 = int(pe.get_action_1d_index_from_spatial_index((r, c, ch)))
                assert i1 == ru.get_action_1d_index_from_spatial_index((r, c, ch))
                for player in (1, -1):
                    assert int(pe.get_action_1d_index_from_player_perspective(i1, player)) == \
                        ru.get_action_1d_index_from_player_perspective(i1, player), (r, c, ch, player)
    for idx in range(-3 * (R + C), ru.action_size + 3 * (R + C)):
        try:
            a = tuple(int(x) for x in pe.get_action_positions_from_1d_index(idx))
        except ValueError:
            a = None
        try:
            b = ru.get_action_positions_from_1d_index(idx)
        except ValueError:
            b = None
        assert a == b, (idx, a, b)


def compare_obs(o_ref, o_orc, where):
    assert sorted(o_ref.keys()) == sorted(o_orc.keys()), (where, o_ref.keys(), o_orc.keys())
    for p in o_ref:
        assert sorted(o_ref[p].keys()) == sorted(o_orc[p].keys()), (where, o_ref[p].keys(), o_orc[p].keys())
        if 'full_observation' in o_ref[p]:
            f_r, f_o = o_ref[p]['full_observation'], o_orc[p]['full_observation']
            assert f_r.dtype == np.float32 and f_r.tobytes() == f_o.tobytes(), (where, 'full obs', p)
        if 'partial_observation' not in o_ref[p]:
            assert np.array_equal(o_ref[p]['valid_actions_mask'], o_orc[p]['valid_actions_mask'])
            continue
        m_r, m_o = o_ref[p]['valid_actions_mask'], o_orc[p]['valid_actions_mask']
        assert m_r.dtype == np.int64 and m_r.shape == m_o.shape
        assert np.array_equal(m_r, m_o), (where, 'mask', p)
        p_r, p_o = o_ref[p]['partial_observation'], o_orc[p]['partial_observation']
        assert p_r.dtype == np.float32 and p_o.dtype == np.float32
        assert p_r.tobytes() == p_o.tobytes(), (where, 'obs', p, np.argwhere(p_r != p_o)[:5])


def play_game(ref, version_name, cfg, rng, garbage_rate, check_fns, mode='partially_observable', channel_mode='extended'):
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    env = ref.maenv.StrategoMultiAgentEnv({'version': GV(version_name), 'observation_mode': OM(mode),
                                           'obs_channel_mode': channel_mode,
                                           'human_inits': version_name in ('standard', 'barrage', 'short_barrage',
                                                                           'medium_standard', 'short_standard')})
    R, C = cfg['rows'], cfg['columns']
    counts = [cfg['piece_amounts'][ref.impl.SP(t)] for t in range(1, 13)]
    oenv = orc.OracleEnv(R, C, cfg['max_turns'], cfg['obstacle_locations'], counts, observation_mode=mode, obs_channel_mode=channel_mode)
    assert np.array_equal(env._p_obs_mids.reshape(-1), oenv.mids) and np.array_equal(env._p_obs_ranges.reshape(-1), oenv.ranges)
    assert np.array_equal(env._f_obs_mids.reshape(-1), oenv.f_mids) and np.array_equal(env._f_obs_ranges.reshape(-1), oenv.f_ranges)
    obs_r = env.reset()
    m1, m2 = own_side_maps_from_state(env.state)
    obs_o = oenv.reset(m1, m2)
    assert np.array_equal(env.state, oenv.state), 'create_initial_state'
    compare_obs(obs_r, obs_o, (version_name, 'reset'))
    K = oenv.K
    NA = R * C * K
    n = 0
    pe, ru = env.base_env, oenv.rules
    while True:
        p = list(obs_r.keys())[0]
        mask = obs_r[p]['valid_actions_mask'].reshape(-1)
        if rng.random() < garbage_rate:
            kind = rng.randrange(5)
            if kind == 0:
                a = rng.randrange(NA)
            elif kind == 1:
                a = rng.randrange(R * C) * K + (K - 1)          # no-op channel at any cell
            elif kind == 2:
                a = rng.choice([-1, NA, NA + 7, -NA])
            elif kind == 3:
                a = (R * C - 1) * K + rng.randrange(K)             # last cell: 1-D index overflows
            else:
                a = rng.randrange(C) * K + rng.randrange(K)        # first row
        else:
            valid = np.flatnonzero(mask)
            a = int(valid[rng.randrange(len(valid))])
        if check_fns and n % 7 == 0:
            st = env.state
            for pl in (1, -1):
                assert np.array_equal(pe.get_valid_moves_as_1d_mask(st, pl), ru.get_valid_moves_as_1d_mask(st, pl))
                assert np.array_equal(pe.get_valid_moves_as_spatial_mask(st, pl), ru.get_valid_moves_as_spatial_mask(st, pl))
                assert np.array_equal(pe.get_state_from_player_perspective(st, pl), ru.get_state_from_player_perspective(st, pl))
                fo_r = pe.get_fully_observable_observation_extended_channels(st, pl)
                assert fo_r.tobytes() == ru.get_fully_observable_observation_extended_channels(st, pl).tobytes()
                assert pe.get_fully_observable_observation(st, pl).tobytes() == ru.get_fully_observable_observation(st, pl).tobytes()
                assert pe.get_partially_observable_observation(st, pl).tobytes() == \
                    ru.get_partially_observable_observation(st, pl).tobytes()
                for idx in [rng.randrange(ru.action_size) for _ in range(8)]:
                    for osc in (False, True):
                        assert bool(pe.is_move_valid_by_1d_index(st, pl, idx, allow_piece_oscillation=osc)) == \
                            ru.is_move_valid_by_1d_index(st, pl, idx, allow_piece_oscillation=osc)
        err_r = err_o = None
        try:
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):   # the reference prints diagnostics on invalid moves
                out_r = env.step({p: a})
        except (ValueError, AssertionError) as e:
            err_r = e
        try:
            out_o = oenv.step({p: a})
        except ValueError as e:
            err_o = e
        assert (err_r is None) == (err_o is None), (version_name, n, a, err_r, err_o)
        assert np.array_equal(env.state, oenv.state) and env.player == oenv.player, (version_name, n, a)
        if err_r is not None:
            continue
        n += 1
        obs_r, rew_r, done_r, info_r = out_r
        obs_o, rew_o, done_o, info_o = out_o
        compare_obs(obs_r, obs_o, (version_name, n))
        assert done_r == done_o, (done_r, done_o)
        assert sorted(rew_r) == sorted(rew_o) and all(float(rew_r[k]) == float(rew_o[k]) for k in rew_r), (rew_r, rew_o)
        assert info_r == info_o, (info_r, info_o)
        if done_r['__all__']:
            # stepping a finished game: both must agree too (error or weird no-op acceptance)
            for a2 in [rng.randrange(NA) for _ in range(6)] + [(R * C - 1) * K + 2 * (R - 1)]:
                e1 = e2 = None
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        o1 = env.step({env.player: a2})
                except (ValueError, AssertionError) as e:
                    e1 = e
                try:
                    o2 = oenv.step({oenv.player: a2})
                except ValueError as e:
                    e2 = e
                assert (e1 is None) == (e2 is None), (version_name, 'post-terminal', a2, e1, e2)
                assert np.array_equal(env.state, oenv.state) and env.player == oenv.player
                if e1 is None:
                    compare_obs(o1[0], o2[0], (version_name, 'post-terminal'))
                    assert o1[2] == o2[2] and o1[3] == o2[3]
            return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games', type=int, default=6)
    args = ap.parse_args()
    ref = import_reference()
    rng = random.Random(1234)
    np.random.seed(99)
    random.seed(99)
    for gv, cfg in ref.maenv.VERSION_CONFIGS.items():
        name = gv.value
        R, C = cfg['rows'], cfg['columns']
        if R * C <= 100:
            check_index_algebra(ref, R, C)
        games = args.games if R * C <= 100 else 1
        if name in ('standard', 'medium_standard'):
            games = max(1, games // 3)
        total = 0
        for g in range(games):
            mode = ('partially_observable', 'both_observations', 'fully_observable')[g % 3]
            total += play_game(ref, name, cfg, rng, garbage_rate=0.15, check_fns=(g == 0), mode=mode)
        # obs_channel_mode='original' (32/33-layer observations, maenv:368-375): one game in BOTH mode
        total += play_game(ref, name, cfg, rng, garbage_rate=0.15, check_fns=False, mode='both_observations',
                           channel_mode='original')
        print("%-16s %d games (+1 original-channel game), %d steps: OK" % (name, games, total), flush=True)
    # Variants the reference has no name for: MORE THAN 8 PIECES OF ONE TYPE (the reference's piece_amounts is unbounded; the
    # normalisation of the captured-count channels then divides by the amount, maenv:288-298) -- the piece sets of
    # tests/test_gpu_many_pieces.py.  The reference resolves a version through VERSION_CONFIGS, so its TINY entry is replaced for the
    # duration of the games (this tool's own process only).
    GV = ref.enums.GameVersions
    SPn = ref.impl.SP
    many = {
        'many66': {'rows': 6, 'columns': 6, 'max_turns': 120, 'obstacle_locations': [],
                   'piece_amounts': {SPn(t): n for t, n in zip(range(1, 13), (0, 9, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1)) if n},
                   'initial_state_usable_rows': 2},
        'many88': {'rows': 8, 'columns': 8, 'max_turns': 160, 'obstacle_locations': [(3, 2), (4, 5)],
                   'piece_amounts': {SPn(t): n for t, n in zip(range(1, 13), (1, 12, 2, 0, 0, 0, 0, 0, 0, 1, 1, 3)) if n},
                   'initial_state_usable_rows': 3},
    }
    saved = ref.maenv.VERSION_CONFIGS[GV.TINY]
    try:
        for name, cfg in many.items():
            cfg = dict(cfg)
            cfg['piece_amounts'] = {SPn(t): cfg['piece_amounts'].get(SPn(t), 0) for t in range(1, 13)}
            ref.maenv.VERSION_CONFIGS[GV.TINY] = cfg
            total = 0
            for g in range(max(3, args.games)):
                mode = ('partially_observable', 'both_observations', 'fully_observable')[g % 3]
                total += play_game(ref, 'tiny', cfg, rng, garbage_rate=0.1, check_fns=(g == 0), mode=mode)
            total += play_game(ref, 'tiny', cfg, rng, garbage_rate=0.1, check_fns=False, mode='both_observations', channel_mode='original')
            print("%-16s (TINY's config entry replaced: up to %d pieces of one type) %d steps: OK" % (name, max(cfg['piece_amounts'].values()), total), flush=True)
    finally:
        ref.maenv.VERSION_CONFIGS[GV.TINY] = saved
    print("oracle == reference on all checks")


if __name__ == '__main__':
    main()
