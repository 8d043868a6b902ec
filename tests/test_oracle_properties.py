"""Property tests (hypothesis) of the CPU restatement and of the product's host-side index algebra / normalisation constants:
size-independent invariants of the domain that hold for the reference by construction."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import oracle as orc
from stratego_env_amd import index_algebra as ia, obs_norm
from stratego_env_amd.config import VARIANTS
from stratego_env_amd.multiagent_env import state_from_player_perspective
from tests.helpers import oracle_cvariant

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow], max_examples=60)
dims = st.tuples(st.integers(3, 15), st.integers(3, 15))


@settings(**COMMON)
@given(dims, st.data())
def test_index_algebra_round_trips_on_any_board(rc, data):
    """positions <-> 1-D <-> spatial are mutually inverse for straight non-null moves, the perspective flip is an involution,
    and the product's numpy index algebra agrees with the oracle (impl:262-396, 678-720) on boards of any size."""
    R, C = rc
    ru = orc.OracleRules(R, C)
    sr, sc = data.draw(st.integers(0, R - 1)), data.draw(st.integers(0, C - 1))
    if data.draw(st.booleans()):
        er, ec = data.draw(st.integers(0, R - 1).filter(lambda x: x != sr)), sc
    else:
        er, ec = sr, data.draw(st.integers(0, C - 1).filter(lambda x: x != sc))
    a1 = ru.get_action_1d_index_from_positions(sr, sc, er, ec)
    assert 0 <= a1 < ru.action_size - 1
    assert ru.get_action_positions_from_1d_index(a1) == (sr, sc, er, ec)
    sp = ru.get_action_spatial_index_from_positions(sr, sc, er, ec)
    assert ru.get_action_positions_from_spatial_index(sp) == (sr, sc, er, ec)
    assert ru.get_action_1d_index_from_spatial_index(sp) == a1
    assert ru.get_action_spatial_index_from_1d_index(a1) == tuple(sp)
    f = ru.get_action_1d_index_from_player_perspective(a1, -1)
    assert ru.get_action_1d_index_from_player_perspective(f, -1) == a1 and ru.get_action_1d_index_from_player_perspective(a1, 1) == a1
    # product host code
    assert int(ia.action_1d_from_positions(R, C, sr, sc, er, ec)) == a1
    assert tuple(int(x) for x in ia.positions_from_1d(R, C, a1)) == (sr, sc, er, ec)
    assert tuple(int(x) for x in ia.spatial_from_positions(R, C, sr, sc, er, ec)) == tuple(sp)
    assert int(ia.action_1d_from_spatial(R, C, *sp)) == a1
    assert int(ia.action_1d_from_player_perspective(R, C, a1, -1)) == f
    assert ia.action_size(R, C) == ru.action_size and ia.spatial_channels(R, C) == ru.K


def _random_position(name, seed, g, n_moves):
    v = VARIANTS[name]
    cv = oracle_cvariant(name)
    ru = orc.OracleRules(v.rows, v.columns)
    state, player = orc.reset_state(cv, seed, g, 0), 1
    rng = np.random.RandomState(seed % 2 ** 31)
    for _ in range(n_moves):
        if ru.get_game_ended(state, player) != 0:
            break
        m = ru.get_valid_moves_as_1d_mask(state, player)
        state, player = ru.get_next_state(state, player, int(rng.choice(np.flatnonzero(m))))
    return v, ru, state, player


positions = st.tuples(st.sampled_from(['tiny', 'fives', 'medium', 'octa_barrage']), st.integers(0, 2 ** 31 - 1),
                      st.integers(0, 1000), st.integers(0, 60))


@settings(**COMMON)
@given(positions)
def test_masks_agree_across_encodings_and_perspectives(pos):
    v, ru, state, player = _random_position(*pos)
    m1 = ru.get_valid_moves_as_1d_mask(state, player)
    ms = ru.get_valid_moves_as_spatial_mask(state, player)
    assert int(m1.sum()) == int(ms.sum()) >= 1 and set(np.unique(m1)) <= {0, 1}
    # every bit of the spatial mask is the same move as a bit of the 1-D mask
    for r, c, ch in np.argwhere(ms):
        if ch == ru.K - 1:
            assert (r, c) == (0, 0) and m1[-1] == 1
        else:
            assert m1[ru.get_action_1d_index_from_spatial_index((r, c, ch))] == 1
    # the mover's perspective: flipping the state and asking as player 1 is the flipped mask
    pers = ru.get_valid_moves_as_spatial_mask(ru.get_state_from_player_perspective(state, player), 1)
    assert int(pers.sum()) == int(ms.sum())
    for r, c, ch in np.argwhere(pers):
        if ch == ru.K - 1:
            continue
        sr, sc, er, ec = ru.get_action_positions_from_spatial_index((r, c, ch))
        if player == -1:
            sr, sc, er, ec = v.rows - 1 - sr, v.columns - 1 - sc, v.rows - 1 - er, v.columns - 1 - ec
        assert m1[ru.get_action_1d_index_from_positions(sr, sc, er, ec)] == 1
    # the perspective flip is an involution, and the product's host-side copy of it agrees
    back = ru.get_state_from_player_perspective(ru.get_state_from_player_perspective(state, -1), -1)
    assert np.array_equal(back, state)
    assert np.array_equal(state_from_player_perspective(state, -1), ru.get_state_from_player_perspective(state, -1))


@settings(**COMMON)
@given(positions)
def test_transitions_conserve_pieces_and_alternate_movers(pos):
    v, ru, state, player = _random_position(*pos)
    if ru.get_game_ended(state, player) != 0:
        return
    m1 = ru.get_valid_moves_as_1d_mask(state, player)
    total = np.asarray(v.piece_counts)
    for a in np.flatnonzero(m1)[:6]:
        assert ru.is_move_valid_by_1d_index(state, player, int(a))
        ns, npl = ru.get_next_state(state, player, int(a))
        assert npl == -player and ns[5, 0, 0] == state[5, 0, 0] + 1
        for pi, cap0 in ((0, 8), (1, 20)):                      # pieces on the board + captured pieces = the variant's set
            on_board = np.bincount(ns[pi].reshape(-1), minlength=13)[1:13]
            captured = ns[cap0:cap0 + 12].reshape(12, -1).sum(axis=1)
            assert np.array_equal(on_board + captured, total), (pi, on_board, captured)
        # what each side knows about the other is either the truth or UNKNOWN, exactly where pieces stand
        for pi in (0, 1):
            known = ns[3 + pi]
            assert np.array_equal(known != 0, ns[pi] != 0)
            assert np.all((known == ns[pi]) | (known == 13))
        assert not np.any((ns[0] != 0) & (ns[1] != 0)) and not np.any(((ns[0] != 0) | (ns[1] != 0)) & (ns[2] != 0))
    invalid = np.flatnonzero(m1 == 0)
    if len(invalid):
        a = int(invalid[len(invalid) // 2])
        assert not ru.is_move_valid_by_1d_index(state, player, a)
        try:
            ru.get_next_state(state, player, a)
            raise AssertionError("an invalid move was accepted")
        except ValueError:
            pass


@settings(**COMMON)
@given(positions, st.booleans())
def test_observations_are_normalised_one_hot_renderings(pos, original):
    v, ru, state, player = _random_position(*pos)
    oe = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts,
                       observation_mode='both_observations', obs_channel_mode='original' if original else 'extended')
    oe.reset(initial_state_override=state, first_player_override=player)
    o = oe._obs(player)
    for key, full in (('partial_observation', False), ('full_observation', True)):
        x = o[key]
        assert x.dtype == np.float32 and np.all(np.isfinite(x)) and float(np.abs(x).max()) <= 1.0
        hi, lo = obs_norm.obs_highs_lows(v.piece_counts, full=full, original=original)
        ranges, mids = obs_norm.ranges_mids(hi, lo)
        raw = x * ranges + mids                                 # denormalize_*_observation (maenv:503-511)
        assert np.allclose(raw, np.round(raw), atol=1e-5)
        assert np.all(np.round(raw) >= lo - 1e-6) and np.all(np.round(raw) <= hi + 1e-6)
        if not original:
            n_true = 24 if full else 12
            onehot = np.round(raw[..., :n_true + 26])
            assert set(np.unique(onehot)) <= {0.0, 1.0}
            assert np.all(onehot[..., :12].sum(axis=-1) <= 1)
