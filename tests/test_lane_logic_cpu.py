"""The lane-per-game rules (stratego_env_amd/csrc/sgx_lane.h: one game per GPU lane on boards of at most 16 cells) against the CPU
oracle, WITHOUT a GPU: tests/lane_harness.hip compiles the very same __host__ __device__ functions for the host
(hipcc --offload-host-only) and this file plays them step by step beside the oracle -- env.step() with valid and garbage actions,
the functional 1-D / position actions with and without the oscillation flag, the k-th valid action, fresh random setups.  Test
infrastructure only: the product never loads the harness."""
import ctypes as C
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'lane_harness.hip')
OUT = os.path.join(ROOT, 'tests', '_build', 'liblane_harness.so')
DEPS = [SRC] + [os.path.join(ROOT, 'stratego_env_amd', 'csrc', f) for f in ('sgx_lane.h', 'sgx_layout.h')] + [os.path.join(ROOT, 'include', 'stratego_mi355x.h')]

ACTIONS_1D, ALLOW_OSC, ACTIONS_POS = 1, 2, 8


@pytest.fixture(scope='module')
def lh():
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available: the host harness of the lane logic cannot be built")
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(f) for f in DEPS):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call([hipcc, '-O2', '-std=c++17', '--offload-host-only', '-fPIC', '-shared', '-fvisibility=hidden', '-Wall',
                               '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'stratego_env_amd', 'csrc'), SRC, '-o', OUT])
    L = C.CDLL(OUT)
    L.lh_step.restype = C.c_int
    L.lh_sample.restype = C.c_int
    L.lh_sample.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p]
    return L


def lane_step(L, R, Cc, state, player, action=0, pos=(0, 0, 0, 0), flags=0, mode=0, max_events=16, k=0):
    K = 2 * (R - 1) + 2 * (Cc - 1) + 1
    st = np.ascontiguousarray(state, dtype=np.int64).copy()
    pl = C.c_int(int(player))
    posa = (C.c_int32 * 4)(*[int(x) for x in pos])
    mask = np.zeros(R * Cc * K, dtype=np.uint8)
    nvalid, kth, oflags = C.c_int(0), C.c_int(0), C.c_int(0)
    rew = (C.c_float * 2)()
    rc = L.lh_step(R, Cc, st.ctypes.data_as(C.c_void_p), C.byref(pl), int(action), posa, int(flags), int(mode), int(max_events), int(k),
                   mask.ctypes.data_as(C.c_void_p), C.byref(nvalid), C.byref(kth), rew, C.byref(oflags))
    assert rc == 0
    return {'state': st, 'player': pl.value, 'mask': mask.reshape(R, Cc, K), 'nvalid': nvalid.value, 'kth': kth.value,
            'rewards': (rew[0], rew[1]), 'invalid': bool(oflags.value & 1), 'done': bool(oflags.value & 2), 'ending_invalid': bool(oflags.value & 4)}


# (rows, cols, pieces per side {type: count}, usable rows, obstacle cells, max_turns)
CONFIGS = {
    'micro': (3, 4, {5: 1, 6: 1, 11: 1}, 1, (), 20),
    'tiny': (4, 4, {5: 1, 6: 1, 7: 1, 11: 1}, 1, (), 100),
    # every rule on 16 cells: scouts (long moves, reveals, the two-square skip), spy / marshal, miner / bomb, a lake
    'zoo44': (4, 4, {1: 1, 2: 2, 3: 1, 10: 1, 11: 1, 12: 1}, 2, ((1, 1),), 60),
    'scouts34': (3, 4, {2: 2, 12: 1, 11: 1}, 1, ((1, 2),), 40),
    'tall43': (4, 3, {2: 1, 3: 1, 12: 1, 11: 1}, 2, (), 50),
    'pairs44': (4, 4, {4: 2, 6: 2}, 1, (), 80),            # no flag at all: games end by capture-out or max turns
}


def _random_maps(rng, R, Cc, pieces, usable, obstacles=()):
    """Own-side piece maps of both players (player -1's is turned by 180 degrees when the state is built, impl:221); no piece on a lake."""
    def one(blocked):
        m = np.zeros((R, Cc), dtype=np.int64)
        cells = [(r, c) for r in range(usable) for c in range(Cc) if (r, c) not in blocked]
        rng.shuffle(cells)
        at = 0
        for t, n in pieces.items():
            for _ in range(n):
                m[cells[at]] = t
                at += 1
        return m
    return one(set(obstacles)), one({(R - 1 - r, Cc - 1 - c) for r, c in obstacles})


def _oracle_env(R, Cc, pieces, obstacles, max_turns):
    counts = [pieces.get(t, 0) for t in range(1, 13)]
    return orc.OracleEnv(R, Cc, max_turns, obstacles, counts)


@pytest.mark.parametrize('name', sorted(CONFIGS))
def test_env_step_matches_the_oracle(lh, name):
    R, Cc, pieces, usable, obstacles, max_turns = CONFIGS[name]
    K = 2 * (R - 1) + 2 * (Cc - 1) + 1
    NA = R * Cc * K
    rng = np.random.RandomState(zlib.crc32(name.encode()) % 1000 + 7)
    env = _oracle_env(R, Cc, pieces, obstacles, max_turns)
    n_steps = n_invalid = n_games = n_capt = n_two_square = 0
    for game in range(250):
        m1, m2 = _random_maps(rng, R, Cc, pieces, usable, obstacles)
        first = env.reset(m1, m2)
        # the observe path on the fresh state: mask of player +1
        ob = lane_step(lh, R, Cc, env.state, 1, mode=1)
        assert np.array_equal(ob['mask'], first[1][env.MASK].astype(np.uint8)) and np.array_equal(ob['state'], env.state)
        n_games += 1
        for t in range(3 * max_turns):
            mask = env._obs(env.player)[env.MASK].astype(np.uint8)
            valid = np.flatnonzero(mask.reshape(-1))
            u = rng.rand()
            if u < 0.80:
                a = int(valid[rng.randint(len(valid))])
            elif u < 0.90:
                a = int(rng.randint(-3, NA + 3))                              # anything, mostly invalid
            else:
                a = int(rng.randint(R * Cc)) * K + (K - 1)                   # the spatial no-op channel of some cell (garbage chain)
            state_before, player_before = env.state.copy(), env.player
            # the k-th valid action of the position BEFORE the step, through the observe path
            if t % 5 == 0:
                k = int(rng.randint(len(valid)))
                o = lane_step(lh, R, Cc, state_before, player_before, mode=1, k=k)
                assert o['kth'] == int(valid[k]) and o['nvalid'] == (0 if (len(valid) == 1 and valid[0] == K - 1) else len(valid))
            got = lane_step(lh, R, Cc, state_before, player_before, action=a)
            try:
                obs, rew, dones, infos = env.step({player_before: a})
                err = False
            except ValueError:
                err = True
            n_steps += 1
            assert got['invalid'] == err, (name, game, t, a)
            if err:
                n_invalid += 1
                assert np.array_equal(got['state'], state_before) and got['player'] == player_before
                continue
            assert np.array_equal(got['state'], env.state), (name, game, t, a, np.argwhere(got['state'] != env.state)[:5])
            n_capt += int(env.state[8:32].sum() > state_before[8:32].sum())
            n_two_square += int((env.state[6:8] == -3).any())
            assert got['done'] == bool(dones['__all__'])
            if dones['__all__']:
                assert got['ending_invalid'] == bool(infos[1]['game_result_was_invalid'])
                assert got['rewards'] == (np.float32(rew[1]), np.float32(rew[-1]))
                assert got['nvalid'] == 0 and got['mask'].sum() == 1 and got['mask'][0, 0, K - 1] == 1
                break
            assert got['player'] == env.player and got['rewards'] == (0.0, 0.0)
            assert np.array_equal(got['mask'], obs[env.player][env.MASK].astype(np.uint8)), (name, game, t)
    assert n_steps > 1000 and n_invalid > 50 and n_capt > 20, (n_steps, n_invalid, n_capt)
    if name in ('tiny', 'zoo44', 'pairs44'):
        assert n_two_square > 0


@pytest.mark.parametrize('name', ['zoo44', 'scouts34', 'micro'])
def test_functional_actions_match_the_oracle(lh, name):
    """1-D actions in absolute coordinates and (sr, sc, er, ec) positions, with and without allow_piece_oscillation
    (penv:87-99, 148-155), on positions reached by random play."""
    R, Cc, pieces, usable, obstacles, max_turns = CONFIGS[name]
    rules = orc.OracleRules(R, Cc)
    AS = rules.action_size
    rng = np.random.RandomState(11)
    env = _oracle_env(R, Cc, pieces, obstacles, max_turns)
    checked = osc_differs = 0
    for game in range(25):
        env.reset(*_random_maps(rng, R, Cc, pieces, usable, obstacles))
        for t in range(max_turns):
            st, pl = env.state.copy(), env.player
            for _ in range(6):
                a1 = int(rng.randint(AS))
                osc = bool(rng.randint(2))
                got = lane_step(lh, R, Cc, st, pl, action=a1, flags=ACTIONS_1D | (ALLOW_OSC if osc else 0))
                try:
                    want, npl = rules.get_next_state(st, pl, a1, allow_piece_oscillation=osc)
                    assert not got['invalid'] and np.array_equal(got['state'], want) and got['player'] == npl, (name, a1, osc)
                except ValueError:
                    assert got['invalid'], (name, a1, osc)
                if rules.is_move_valid_by_1d_index(st, pl, a1, True) != rules.is_move_valid_by_1d_index(st, pl, a1, False):
                    osc_differs += 1
                p = [int(x) for x in rng.randint(-1, max(R, Cc) + 1, size=4)]
                gotp = lane_step(lh, R, Cc, st, pl, pos=p, flags=ACTIONS_POS)
                assert gotp['invalid'] == (not rules.is_move_valid_by_position(st, pl, *p)), (name, p)
                checked += 2
            mask = env._obs(env.player)[env.MASK]
            valid = np.flatnonzero(mask.reshape(-1))
            _, _, dones, _ = env.step({env.player: int(valid[rng.randint(len(valid))])})
            if dones['__all__']:
                break
    assert checked > 1000 and osc_differs >= 0


def test_fourth_oscillation_with_and_without_the_flag(lh):
    """SURVEY A.5's two-square timeline on 4x4: after X->Y, Y->X, X->Y the move Y->X is illegal for the mask and for get_next_state,
    legal for get_next_state(allow_piece_oscillation=True); the mask keeps the veto either way (impl:439-445, 771-777)."""
    R = Cc = 4
    rules = orc.OracleRules(R, Cc)
    st = np.zeros((34, R, Cc), dtype=np.int64)
    st[5, 1, 0] = 100
    st[0, 0, 0] = 5; st[3, 0, 0] = 13; st[32, 0, 0] = 1        # player +1: a lieutenant that will oscillate (0,0) <-> (1,0), and a flag
    st[0, 0, 3] = 11; st[3, 0, 3] = 13; st[32, 0, 3] = 1
    st[1, 3, 3] = 6; st[4, 3, 3] = 13; st[33, 3, 3] = 1        # player -1: a captain shuffling (3,3) <-> (3,2), and a flag
    st[1, 3, 0] = 11; st[4, 3, 0] = 13; st[33, 3, 0] = 1
    pl = 1
    a_down, a_up = rules.get_action_1d_index_from_positions(0, 0, 1, 0), rules.get_action_1d_index_from_positions(1, 0, 0, 0)
    b_left, b_right = rules.get_action_1d_index_from_positions(3, 3, 3, 2), rules.get_action_1d_index_from_positions(3, 2, 3, 3)
    for a in (a_down, b_left, a_up, b_right, a_down, b_left):
        got = lane_step(lh, R, Cc, st, pl, action=a, flags=ACTIONS_1D)
        want, npl = rules.get_next_state(st, pl, a)
        assert not got['invalid'] and np.array_equal(got['state'], want) and got['player'] == npl
        st, pl = want, npl
    assert pl == 1 and st[6, 1, 0] == -3 and st[6, 0, 0] == 1
    assert not rules.is_move_valid_by_1d_index(st, 1, a_up, False) and rules.is_move_valid_by_1d_index(st, 1, a_up, True)
    assert lane_step(lh, R, Cc, st, 1, action=a_up, flags=ACTIONS_1D)['invalid']
    got = lane_step(lh, R, Cc, st, 1, action=a_up, flags=ACTIONS_1D | ALLOW_OSC)
    want, _ = rules.get_next_state(st, 1, a_up, allow_piece_oscillation=True)
    assert not got['invalid'] and np.array_equal(got['state'], want)
    ob = lane_step(lh, R, Cc, st, 1, mode=1)
    K = 2 * (R - 1) + 2 * (Cc - 1) + 1
    assert ob['mask'].reshape(R, Cc, K)[1, 0].tolist() == [1, 0, 0, 0, 0, 0, 1] + [0] * 6      # down 1 and right 1; up 1 is vetoed


@pytest.mark.parametrize('name', ['micro', 'tiny', 'zoo44'])
def test_fresh_setups_match_the_oracle(lh, name):
    """Fisher-Yates of the usable back cells with the counter RNG keyed by (seed, global env id, game): the oracle's so_reset_env."""
    R, Cc, pieces, usable, obstacles, max_turns = CONFIGS[name]
    counts = [pieces.get(t, 0) for t in range(1, 13)]
    cv = orc.make_cvariant(R, Cc, max_turns, obstacles, counts, usable)
    obst = sum(1 << (r * Cc + c) for r, c in obstacles)
    pc = (C.c_int32 * 12)(*counts)
    for seed, g, j in [(1, 0, 0), (0x5712A7E60, 65535, 3), (99, 123456789, 41), (7, 5, 1), (2 ** 63 + 5, 2 ** 40, 1000)]:
        st = np.zeros((34, R, Cc), dtype=np.int64)
        assert lh.lh_sample(R, Cc, pc, usable, seed, g, j, max_turns, obst, st.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(st, orc.reset_state(cv, seed, g, j)), (name, seed, g, j)
