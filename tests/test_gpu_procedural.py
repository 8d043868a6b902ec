"""N2: the batched functional StrategoProceduralEnv API on caller-provided states vs the oracle's OracleRules (which
tools/oracle/check_oracle_vs_reference.py pins against the reference's penv methods, oscillation flag included)."""
import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd.config import VARIANTS
from tests.helpers import load_games, oracle_env

pytestmark = pytest.mark.gpu


def _sample_states(name, n, rs):
    """states / players sampled along golden games (including terminal ones)."""
    g = load_games(name)
    off = g['offsets']
    env = oracle_env(name)
    states, players = [], []
    gi = 0
    while len(states) < n:
        env.reset(g['p1_maps'][gi].astype(np.int64), g['p2_maps'][gi].astype(np.int64))
        ks = [k for k in range(off[gi], off[gi + 1]) if not g['errors'][k]]
        take = set(rs.choice(len(ks), size=min(len(ks), 12), replace=False).tolist()) | {len(ks) - 1}
        for i, k in enumerate(ks):
            env.step({env.player: int(g['actions'][k])})
            if i in take and len(states) < n:
                states.append(env.state.copy()); players.append(env.player if i != len(ks) - 1 else int(rs.choice([1, -1])))
        gi = (gi + 1) % (len(off) - 1)
    return np.stack(states), np.asarray(players, dtype=np.int8)


@pytest.mark.parametrize('name', ['barrage', 'standard', 'tiny', 'fives', 'octa_barrage'])
def test_functional_api_matches_oracle(name):
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    v = VARIANTS[name]
    rs = np.random.RandomState(11)
    n = 96
    states, players = _sample_states(name, n, rs)
    ru = orc.OracleRules(v.rows, v.columns)
    pe = BatchedStrategoProceduralEnv(name, n)
    assert pe.action_size == ru.action_size and tuple(pe.spatial_action_size) == ru.spatial_action_size
    # masks, perspective, observations, game-ended for BOTH players on every state
    for pl_all in (players, -players):
        m1 = pe.get_valid_moves_as_1d_mask(states, pl_all).cpu().numpy()
        ms = pe.get_valid_moves_as_spatial_mask(states, pl_all).cpu().numpy()
        pp = pe.get_state_from_player_perspective(states, pl_all).cpu().numpy()
        po = pe.get_partially_observable_observation_extended_channels(states, pl_all).cpu().numpy()
        fo = pe.get_fully_observable_observation_extended_channels(states, pl_all).cpu().numpy()
        ge = pe.get_game_ended(states, pl_all).cpu().numpy()
        gi = pe.get_game_result_is_invalid(states).cpu().numpy()
        for e in range(n):
            p = int(pl_all[e])
            assert np.array_equal(m1[e], ru.get_valid_moves_as_1d_mask(states[e], p)), (name, e, '1d mask')
            assert np.array_equal(ms[e], ru.get_valid_moves_as_spatial_mask(states[e], p)), (name, e, 'spatial mask')
            assert np.array_equal(pp[e], ru.get_state_from_player_perspective(states[e], p))
            assert po[e].tobytes() == ru.get_partially_observable_observation_extended_channels(states[e], p).tobytes()
            assert fo[e].tobytes() == ru.get_fully_observable_observation_extended_channels(states[e], p).tobytes()
            assert np.float32(ge[e]) == np.float32(ru.get_game_ended(states[e], p))
            assert bool(gi[e]) == ru.get_game_result_is_invalid(states[e])
    # transitions: valid moves, garbage indices, the 1-D no-op, with and without oscillation
    for rnd in range(6):
        acts = np.zeros(n, dtype=np.int64)
        for e in range(n):
            mask = ru.get_valid_moves_as_1d_mask(states[e], int(players[e]))
            kind = rs.randint(4)
            if kind < 2:
                acts[e] = rs.choice(np.flatnonzero(mask))
            elif kind == 2:
                acts[e] = rs.randint(-5, ru.action_size + 5)
            else:
                acts[e] = ru.action_size - 1
        for osc in (False, True):
            ns, npl, valid = pe.get_next_state(states, players, acts, allow_piece_oscillation=osc)
            ns, npl, valid = ns.cpu().numpy(), npl.cpu().numpy(), valid.cpu().numpy()
            v2 = pe.is_move_valid_by_1d_index(states, players, acts, allow_piece_oscillation=osc).cpu().numpy()
            for e in range(n):
                want_valid = ru.is_move_valid_by_1d_index(states[e], int(players[e]), int(acts[e]), allow_piece_oscillation=osc)
                assert bool(valid[e]) == want_valid == bool(v2[e]), (name, rnd, e, acts[e], osc)
                if want_valid:
                    w, wp = ru.get_next_state(states[e], int(players[e]), int(acts[e]), allow_piece_oscillation=osc)
                    assert np.array_equal(ns[e], w) and npl[e] == wp, (name, rnd, e, 'next state')
                else:
                    assert np.array_equal(ns[e], states[e]) and npl[e] == players[e]
    # validity by raw positions (diagonals, out-of-board, long moves)
    pos = rs.randint(-1, max(v.rows, v.columns) + 1, size=(n, 4))
    for osc in (False, True):
        got = pe.is_move_valid_by_position(states, players, pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3], allow_piece_oscillation=osc).cpu().numpy()
        for e in range(n):
            assert bool(got[e]) == ru.is_move_valid_by_position(states[e], int(players[e]), *[int(x) for x in pos[e]], allow_piece_oscillation=osc)
    pe.close()


def test_two_square_oscillation_flag():
    """The 4th oscillation is illegal (SURVEY 8c known answer) unless allow_piece_oscillation=True (impl:771-777)."""
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    ru = orc.OracleRules(4, 4)
    m1 = np.zeros((4, 4), dtype=np.int64); m2 = np.zeros((4, 4), dtype=np.int64)
    m1[0, 0] = 5; m1[0, 3] = 11; m2[0, 0] = 5; m2[0, 3] = 11
    st = ru.create_initial_state(np.zeros((4, 4), dtype=np.int64), m1, m2, 100)
    pl = 1
    for (s, e) in [((0, 0), (1, 0)), ((3, 3), (2, 3)), ((1, 0), (0, 0)), ((2, 3), (3, 3)), ((0, 0), (1, 0)), ((3, 3), (2, 3))]:
        st, pl = ru.get_next_state(st, pl, ru.get_action_1d_index_from_positions(*s, *e))
    pe = BatchedStrategoProceduralEnv('tiny', 2)
    states = np.stack([st, st]); players = np.asarray([1, 1], dtype=np.int8)
    idx = ru.get_action_1d_index_from_positions(1, 0, 0, 0)
    assert pe.is_move_valid_by_1d_index(states, players, [idx, idx]).cpu().numpy().tolist() == [False, False]
    assert pe.is_move_valid_by_1d_index(states, players, [idx, idx], allow_piece_oscillation=True).cpu().numpy().tolist() == [True, True]
    pe.close()
