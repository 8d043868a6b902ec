"""Board sizes the reference has no variant for.  StrategoProceduralEnv(rows, columns) takes ANY size >= 3 (penv:27-36); here the
kernels of a size that is not compiled into the main library are built on first use (stratego_env_amd/build.py build_geometry:
the same sources, one more template instantiation).  Parity vs the oracle in step mode (every output of every step of
random-valid-action rollouts with auto-reset and garbage actions) and in functional mode (masks, observations, transitions on
sampled states), through the batched API and through the reference-shaped single-state class."""
import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd import config
from stratego_env_amd.config import custom_variant
from tests.helpers import oracle_cvariant, oracle_env

pytestmark = pytest.mark.gpu

#                                                        spy scout miner sgt lt cpt maj col gen mar flag bomb
CUSTOM = {
    'c3x3': custom_variant(3, 3, max_turns=24, piece_counts=(0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0), name='c3x3'),
    'c7x7': custom_variant(7, 7, max_turns=150, obstacle_locations=((3, 1), (3, 5)),
                           piece_counts=(1, 2, 1, 0, 1, 1, 0, 1, 0, 1, 1, 2), initial_state_usable_rows=2, name='c7x7'),
    'c9x5': custom_variant(9, 5, max_turns=200, obstacle_locations=((4, 2),),
                           piece_counts=(1, 3, 2, 1, 1, 1, 1, 0, 1, 1, 1, 2), initial_state_usable_rows=3, name='c9x5'),
    'c12x12': custom_variant(12, 12, max_turns=300, obstacle_locations=((5, 2), (6, 2), (5, 3), (6, 3), (5, 8), (6, 8), (5, 9), (6, 9)),
                             piece_counts=(1, 8, 5, 4, 4, 4, 3, 2, 1, 1, 1, 6), initial_state_usable_rows=4, name='c12x12'),
    'c3x40': custom_variant(3, 40, max_turns=120, piece_counts=(1, 6, 2, 1, 1, 1, 1, 1, 1, 1, 1, 3), name='c3x40'),   # K = 83 > 64 lanes
    # more than 256 cells: 10-bit cell indices in the packed record (uint32 capture events, 6-bit recent-move codes), templates read from
    # global memory, records staged in a loop
    'c20x20': custom_variant(20, 20, max_turns=400, obstacle_locations=((9, 4), (10, 4), (9, 5), (10, 5), (9, 14), (10, 14), (9, 15), (10, 15)),
                             piece_counts=(1, 8, 5, 4, 4, 4, 3, 2, 1, 1, 1, 6), initial_state_usable_rows=3, name='c20x20'),
    'c32x32': custom_variant(32, 32, max_turns=250, obstacle_locations=((15, 7), (16, 7), (15, 24), (16, 24)),      # SGX_MAX_CELLS
                             piece_counts=(1, 8, 5, 4, 4, 4, 3, 2, 1, 1, 1, 6), initial_state_usable_rows=2, name='c32x32'),
    'c17x16': custom_variant(17, 16, max_turns=300, obstacle_locations=((8, 3), (8, 12)),
                             piece_counts=(1, 6, 3, 2, 2, 2, 2, 1, 1, 1, 1, 4), initial_state_usable_rows=2, name='c17x16'),
}


@pytest.fixture(autouse=True)
def _custom_names():
    """The shared checkers look variants up by name: register the test-only names for the duration of a test (the product
    takes the Variant objects themselves)."""
    config.VARIANTS.update(CUSTOM)
    yield
    for k in CUSTOM:
        config.VARIANTS.pop(k, None)


@pytest.mark.parametrize('name,n_envs,n_steps', [('c3x3', 48, 80), ('c7x7', 32, 200), ('c9x5', 32, 200), ('c12x12', 12, 250), ('c3x40', 12, 120),
                                                 ('c20x20', 6, 300), ('c17x16', 8, 250), ('c32x32', 3, 160)])
def test_step_mode_bit_exact_vs_oracle(name, n_envs, n_steps):
    from tests.test_gpu_parity import test_step_bit_exact_vs_oracle
    test_step_bit_exact_vs_oracle(name, n_envs, n_steps, 0.1)


def _sampled_states(name, n, seed):
    """states / movers along oracle rollouts from random setups (terminal states included)."""
    rs = np.random.RandomState(seed)
    cv = oracle_cvariant(name)
    states, players = [], []
    g = 0
    while len(states) < n:
        env = oracle_env(name)
        obs = env.reset(initial_state_override=orc.reset_state(cv, 77 + seed, g, 0))
        g += 1
        for t in range(60):
            p = env.player
            mask = obs[p]['valid_actions_mask']
            a = int(rs.choice(np.flatnonzero(mask.reshape(-1))))
            obs, rew, done, info = env.step({p: a})
            if rs.rand() < 0.35 or done['__all__']:
                states.append(env.state.copy())
                players.append(env.player if not done['__all__'] else int(rs.choice([1, -1])))
            if done['__all__'] or len(states) >= n:
                break
    return np.stack(states[:n]), np.asarray(players[:n], dtype=np.int8)


@pytest.mark.parametrize('name', ['c3x3', 'c7x7', 'c9x5', 'c12x12', 'c3x40', 'c20x20', 'c17x16', 'c32x32'])
def test_functional_mode_matches_oracle(name):
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    v = CUSTOM[name]
    n = 64 if v.cells <= 400 else 12
    states, players = _sampled_states(name, n, 5)
    ru = orc.OracleRules(v.rows, v.columns)
    pe = BatchedStrategoProceduralEnv(v, n)
    assert pe.action_size == ru.action_size and tuple(pe.spatial_action_size) == ru.spatial_action_size
    rs = np.random.RandomState(3)
    for pl_all in (players, -players):
        m1 = pe.get_valid_moves_as_1d_mask(states, pl_all).cpu().numpy()
        assert int(pe.last_sanitised.sum()) == 0
        ms = pe.get_valid_moves_as_spatial_mask(states, pl_all).cpu().numpy()
        po = pe.get_partially_observable_observation_extended_channels(states, pl_all).cpu().numpy()
        fo = pe.get_fully_observable_observation_extended_channels(states, pl_all).cpu().numpy()
        acts = np.zeros(n, dtype=np.int64)
        for e in range(n):
            p = int(pl_all[e])
            want = ru.get_valid_moves_as_1d_mask(states[e], p)
            assert np.array_equal(m1[e], want), (name, e, '1d mask')
            assert np.array_equal(ms[e], ru.get_valid_moves_as_spatial_mask(states[e], p)), (name, e, 'spatial mask')
            assert po[e].tobytes() == ru.get_partially_observable_observation_extended_channels(states[e], p).tobytes(), (name, e)
            assert fo[e].tobytes() == ru.get_fully_observable_observation_extended_channels(states[e], p).tobytes(), (name, e)
            acts[e] = rs.choice(np.flatnonzero(want)) if rs.rand() < 0.8 else rs.randint(ru.action_size)
        ns, npl, ok = pe.get_next_state(states, pl_all, acts)
        ns, npl, ok = ns.cpu().numpy(), npl.cpu().numpy(), ok.cpu().numpy()
        valid = pe.is_move_valid_by_1d_index(states, pl_all, acts).cpu().numpy()
        for e in range(n):
            p = int(pl_all[e])
            good = ru.is_move_valid_by_1d_index(states[e], p, int(acts[e]))
            assert bool(ok[e]) == bool(valid[e]) == bool(good), (name, e, int(acts[e]))
            if good:
                assert np.array_equal(ns[e], ru.get_next_state(states[e], p, int(acts[e]))[0]), (name, e, 'next state')
                assert npl[e] == -p
            else:
                assert np.array_equal(ns[e], states[e]) and npl[e] == p
    pe.close()


def test_reference_shaped_class_takes_any_size_and_any_obstacles():
    """StrategoProceduralEnv(rows, columns) like penv:27-36: no variant name, obstacle cells read from the states."""
    from stratego_env_amd.procedural_env import StrategoProceduralEnv
    with pytest.raises(ValueError):
        StrategoProceduralEnv(2, 9)
    with pytest.raises(ValueError):
        StrategoProceduralEnv(20, 20)                     # documented limit of this build: rows * columns <= 256
    for (R, C), obstacles in (((7, 7), ((3, 3),)), ((6, 9), ()), ((10, 10), ((4, 4), (5, 5)))):      # 10x10 with NON-variant lakes
        env = StrategoProceduralEnv(R, C)
        ru = orc.OracleRules(R, C)
        assert env.action_size == ru.action_size and tuple(int(x) for x in env.spatial_action_size) == ru.spatial_action_size
        ob = np.zeros((R, C), dtype=np.int64)
        for r, c in obstacles:
            ob[r, c] = 1
        m1 = np.zeros((R, C), dtype=np.int64)
        m1[0, :4] = [11, 2, 5, 12]
        m1[1, 1] = 3
        m2 = np.zeros((R, C), dtype=np.int64)
        m2[0, :3] = [11, 2, 9]
        m2[1, 2] = 10
        st = env.create_initial_state(ob, m1, m2, 40)
        assert np.array_equal(st, ru.create_initial_state(ob, m1, m2, 40))
        player = 1
        rs = np.random.RandomState(R * 31 + C)
        for t in range(40):
            mask = env.get_valid_moves_as_1d_mask(st, player)
            assert np.array_equal(mask, ru.get_valid_moves_as_1d_mask(st, player)), (R, C, t)
            assert env.get_partially_observable_observation_extended_channels(st, player).tobytes() == \
                ru.get_partially_observable_observation_extended_channels(st, player).tobytes()
            if env.get_game_ended(st, player) != 0:
                break
            a = int(rs.choice(np.flatnonzero(mask)))
            st2, p2 = env.get_next_state(st, player, a)
            want, _ = ru.get_next_state(st, player, a)
            assert np.array_equal(st2, want) and p2 == -player
            st, player = st2, p2
        env.close()


def test_long_scout_moves_on_a_3x85_board_check_every_intermediate_cell():
    """A board side of more than 66 cells: the scout path check walks the intermediate cells in slices of 64 lanes (it used to look
    at the first 64 only).  Directed positions on 3 x 85: a scout at (1, 2) moving along the row with a blocker 3 .. 80 cells away,
    in front of and behind the target -- is_move_valid / get_next_state / masks vs the oracle."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    R, C = 3, 85
    v = custom_variant(R, C, max_turns=500, piece_counts=(0, 2, 0, 0, 1, 0, 0, 0, 0, 0, 1, 1), name='c3x85')
    ru = orc.OracleRules(R, C)
    states, players, actions = [], [], []
    for mover in (1, -1):
        for blocker_c in (5, 40, 66, 67, 68, 70, 75, 82, None):
            for target_c in (60, 69, 71, 83, 84):
                for blocker_owner in (1, -1):
                    st = np.zeros((34, R, C), dtype=np.int64)
                    st[5, 0, 0], st[5, 1, 0] = 7, 500
                    pi, oi = (0, 1) if mover == 1 else (1, 0)
                    st[pi, 1, 2] = 2; st[3 + pi, 1, 2] = 13; st[32 + pi, 1, 2] = 1           # the scout
                    st[pi, 0, 0] = 11; st[3 + pi, 0, 0] = 13; st[32 + pi, 0, 0] = 1          # flags out of the way
                    st[oi, 2, 84] = 11; st[3 + oi, 2, 84] = 13; st[32 + oi, 2, 84] = 1
                    st[oi, 2, 0] = 5; st[3 + oi, 2, 0] = 13                                  # a spare mover for the opponent
                    if blocker_c is not None:
                        bi = pi if blocker_owner == mover else oi
                        st[bi, 1, blocker_c] = 12; st[3 + bi, 1, blocker_c] = 13; st[32 + bi, 1, blocker_c] = 1
                    states.append(st); players.append(mover)
                    actions.append(ru.get_action_1d_index_from_positions(1, 2, 1, target_c))
    n = len(states)
    states, players, actions = np.stack(states), np.asarray(players, dtype=np.int8), np.asarray(actions, dtype=np.int64)
    pe = BatchedStrategoProceduralEnv(v, n)
    valid = pe.is_move_valid_by_1d_index(states, players, actions).cpu().numpy()
    ns, npl, ok = pe.get_next_state(states, players, actions)
    m1 = pe.get_valid_moves_as_1d_mask(states, players).cpu().numpy()
    ns, ok = ns.cpu().numpy(), ok.cpu().numpy()
    n_valid = 0
    for e in range(n):
        p = int(players[e])
        good = ru.is_move_valid_by_1d_index(states[e], p, int(actions[e]))
        assert bool(valid[e]) == bool(ok[e]) == bool(good), e
        assert np.array_equal(m1[e], ru.get_valid_moves_as_1d_mask(states[e], p)), e
        if good:
            n_valid += 1
            assert np.array_equal(ns[e], ru.get_next_state(states[e], p, int(actions[e]))[0]), e
    assert 0 < n_valid < n
    pe.close()
