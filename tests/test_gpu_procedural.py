"""N2: the batched functional StrategoProceduralEnv API on caller-provided states vs the oracle's OracleRules (which
tools/oracle/check_oracle_vs_reference.py pins against the reference's penv methods, oscillation flag included)."""
import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd.config import VARIANTS
from tests.helpers import general_states, load_games, oracle_cvariant, oracle_env
from tests.test_gpu_parity import _table

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
        # the other observation kinds (79-channel, 'original' channels): three launches on packed records first, then the same second pass
        others = []
        for fn_name in ('get_fully_observable_observation_extended_channels', 'get_partially_observable_observation',
                        'get_fully_observable_observation'):
            others.append((fn_name, getattr(pe, fn_name)(states, pl_all).cpu().numpy()))
            assert int(pe.last_sanitised.sum()) == 0, (name, fn_name)
        for e in range(n):
            p = int(pl_all[e])
            for fn_name, got in others:
                assert got[e].tobytes() == getattr(ru, fn_name)(states[e], p).tobytes(), (name, e, fn_name)
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


@pytest.mark.parametrize('name', ['barrage', 'micro', 'fives'])
def test_single_state_class_has_the_reference_interface(name):
    """stratego_env_amd.procedural_env.StrategoProceduralEnv(rows, columns): the reference's class (penv:20-214) for one state
    at a time -- same calls, return types and errors -- checked against OracleRules on positions from random play."""
    from stratego_env_amd.procedural_env import StrategoProceduralEnv
    v = VARIANTS[name]
    R, C = v.rows, v.columns
    pe = StrategoProceduralEnv(R, C, version=name)
    ru = orc.OracleRules(R, C)
    assert int(pe.action_size) == ru.action_size and tuple(int(x) for x in pe.spatial_action_size) == ru.spatial_action_size
    with pytest.raises(ValueError):
        StrategoProceduralEnv(2, 5)
    rng = np.random.RandomState(11)
    cv = oracle_cvariant(name, setups=_table(name))
    st, pl = orc.reset_state(cv, 5, 0, 0), 1
    # create_initial_state from the own-side maps of that state
    m1, m2 = st[0].copy(), st[1][::-1, ::-1].copy()
    made = pe.create_initial_state(v.obstacle_map().astype(np.int64), m1, m2, v.max_turns)
    assert made.dtype == np.int64 and np.array_equal(made, st)
    with pytest.raises(ValueError):
        pe.create_initial_state(np.zeros((R + 1, C), dtype=np.int64), m1, m2, v.max_turns)
    for t in range(20):
        if ru.get_game_ended(st, pl) != 0:
            break
        m1d = pe.get_valid_moves_as_1d_mask(st, pl)
        assert m1d.dtype == np.int64 and np.array_equal(m1d, ru.get_valid_moves_as_1d_mask(st, pl))
        ms = pe.get_valid_moves_as_spatial_mask(st, pl)
        assert ms.dtype == np.int64 and np.array_equal(ms, ru.get_valid_moves_as_spatial_mask(st, pl))
        assert np.array_equal(pe.get_state_from_player_perspective(st, pl), ru.get_state_from_player_perspective(st, pl))
        po = pe.get_partially_observable_observation_extended_channels(st, pl)
        assert po.dtype == np.float32 and po.tobytes() == ru.get_partially_observable_observation_extended_channels(st, pl).tobytes()
        assert pe.get_fully_observable_observation(st, pl).tobytes() == ru.get_fully_observable_observation(st, pl).tobytes()
        assert float(pe.get_game_ended(st, pl)) == ru.get_game_ended(st, pl)
        assert pe.get_game_result_is_invalid(st) == ru.get_game_result_is_invalid(st)
        a = int(rng.choice(np.flatnonzero(m1d)))
        if a != ru.action_size - 1:
            pos = pe.get_action_positions_from_1d_index(a)
            assert tuple(int(x) for x in pos) == ru.get_action_positions_from_1d_index(a)
            assert int(pe.get_action_1d_index_from_positions(*pos)) == a
            sp = pe.get_action_spatial_index_from_1d_index(a)
            assert tuple(int(x) for x in sp) == ru.get_action_spatial_index_from_1d_index(a)
            assert int(pe.get_action_1d_index_from_spatial_index(sp)) == a
            assert tuple(int(x) for x in pe.get_action_positions_from_spatial_index(sp)) == tuple(int(x) for x in pos)
            assert pe.is_move_valid_by_position(st, pl, *pos) is True
            assert int(pe.get_action_1d_index_from_player_perspective(a, -1)) == ru.get_action_1d_index_from_player_perspective(a, -1)
        assert pe.is_move_valid_by_1d_index(st, pl, a) is True
        bad = int(rng.choice(np.flatnonzero(m1d == 0)))
        assert pe.is_move_valid_by_1d_index(st, pl, bad) is False
        with pytest.raises(ValueError):
            pe.get_next_state(st, pl, bad)
        ns, npl = pe.get_next_state(st, pl, a)
        want, wpl = ru.get_next_state(st, pl, a)
        assert ns.dtype == np.int64 and np.array_equal(ns, want) and npl == wpl == -pl
        st, pl = ns, npl
    with pytest.raises(ValueError):
        pe.get_action_positions_from_1d_index(ru.action_size - 1)
    pe.close()


def test_facade_exposes_base_env():
    from stratego_env_amd import GameVersions, ObservationModes
    from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv
    env = StrategoMultiAgentEnv({'version': GameVersions.TINY, 'observation_mode': ObservationModes.PARTIALLY_OBSERVABLE})
    obs = env.reset()
    be = env.base_env
    assert tuple(int(x) for x in be.spatial_action_size) == tuple(env.spatial_action_size)
    pp = be.get_state_from_player_perspective(env.state, env.player)
    mask = be.get_valid_moves_as_spatial_mask(pp, 1)                                    # maenv:452-454
    assert np.array_equal(mask, obs[env.player]['valid_actions_mask'])
    env.close()


def test_every_call_imports_and_the_loaded_scope_is_opt_in():
    """Default: every query imports the states it is given, so writes that torch cannot see (another library writing through
    data_ptr, `.data` assignments) are never missed.  `with penv.loaded(states, players)` is the opt-in to import once."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    from stratego_env_amd.vec_env import VecStrategoEnv
    n = 64
    env = VecStrategoEnv('barrage', n, seed=12, auto_reset=True)
    env.reset()
    env.rollout_steps(40)
    states, players = env.export_state()
    penv = BatchedStrategoProceduralEnv('barrage', n)
    fresh = BatchedStrategoProceduralEnv('barrage', n)
    m1 = penv.get_valid_moves_as_1d_mask(states, players)
    # a write that does not bump torch's version counter (what a foreign kernel writing through data_ptr() looks like)
    env.rollout_steps(7)
    s2, p2 = env.export_state()
    v0 = states._version
    states.data.copy_(s2); players.data.copy_(p2)
    assert states._version == v0
    m3 = penv.get_valid_moves_as_1d_mask(states, players)
    assert torch.equal(m3, fresh.get_valid_moves_as_1d_mask(s2, p2)) and not torch.equal(m3, m1)
    # opt-in scope: one import for several questions about the same objects; a transition inside the scope re-imports
    calls = []
    class Counting:                                                       # (both entry points that import int64 states)
        def __getattr__(self, k):
            if k in ('sgx_import_state_checked', 'sgx_step_states'):
                real = getattr(penv_L, k)
                return lambda *a: (calls.append(1), real(*a))[1]
            return getattr(penv_L, k)
    penv_L = penv._vec._L
    penv._vec._L = Counting()
    with penv.loaded(states, players):
        ma = penv.get_valid_moves_as_1d_mask(states, players)
        po = penv.get_partially_observable_observation_extended_channels(states, players)
        assert len(calls) == 1 and torch.equal(ma, m3)
        a = torch.argmax((m3 != 0).to(torch.int8), dim=1)
        penv.get_next_state(states, players, a)                           # uses the held import, then dirties the scratch handle
        assert len(calls) == 1
        assert torch.equal(penv.get_valid_moves_as_1d_mask(states, players), m3) and len(calls) == 2
        other = states.clone()
        penv.get_valid_moves_as_1d_mask(other, players)                   # other objects: imported, and the held ones again afterwards
        assert torch.equal(penv.get_valid_moves_as_1d_mask(states, players), m3) and len(calls) == 4
    penv.get_valid_moves_as_1d_mask(states, players)
    penv.get_valid_moves_as_1d_mask(states, players)
    assert len(calls) == 6                                                # outside the scope: always
    penv._vec._L = penv_L
    assert torch.equal(po, fresh.get_partially_observable_observation_extended_channels(s2, p2))
    env.close(); penv.close(); fresh.close()


def test_import_reports_sanitised_states_and_strict_mode_raises():
    """The packed record carries reachable states only; sgx_import_state_checked flags every state it had to alter, the
    batched API exposes the flags and the single-state class raises ValueError."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv, StrategoProceduralEnv
    from stratego_env_amd.vec_env import VecStrategoEnv
    n = 16
    env = VecStrategoEnv('barrage', n, seed=3, auto_reset=True)
    env.reset()
    env.rollout_steps(30)
    states, players = env.export_state()
    bad = states.clone()
    bad[1, 0, 0, 0] = 99                 # piece code out of range
    bad[2, 6, 3, 3] = 1; bad[2, 6, 3, 4] = -1; bad[2, 6, 3, 5] = -2      # three recent-move cells of one player
    bad[3, 2, 0, 0] = 1                  # an obstacle the variant does not have
    bad[4, 10] = 3                       # 300 captured pieces
    bad[5, 33, 2, 2] = 7                 # never-moved flag that is not 0 / 1
    penv = BatchedStrategoProceduralEnv('barrage', n)
    m = penv.get_valid_moves_as_1d_mask(bad, players).cpu().numpy()
    flags = penv.last_sanitised.cpu().numpy()
    # round 4: the one-call functional API redoes states 2 and 4 on the general-state kernels (legal values, impossible structure) -- they
    # come back exact and unflagged; values outside a layer's range and a foreign obstacle stay flagged
    assert flags.tolist() == [0, 1, 0, 1, 0, 1] + [0] * (n - 6)
    ru = orc.OracleRules(10, 10)
    for e in (2, 4):
        assert np.array_equal(m[e], ru.get_valid_moves_as_1d_mask(bad[e].cpu().numpy(), int(players[e])))
    penv.get_valid_moves_as_1d_mask(states, players)
    assert int(penv.last_sanitised.sum()) == 0
    packed = penv.pack(bad, players)            # packed records (search pools) still carry reachable states only: all five flagged
    assert packed.sanitised.cpu().numpy().tolist() == [0, 1, 1, 1, 1, 1] + [0] * (n - 6)
    penv.strict = True
    with pytest.raises(ValueError):
        penv.get_valid_moves_as_1d_mask(bad, players)
    single = StrategoProceduralEnv(10, 10)
    single.get_valid_moves_as_1d_mask(states[0].cpu().numpy(), int(players[0]))
    with pytest.raises(ValueError):
        single.get_valid_moves_as_1d_mask(bad[1].cpu().numpy(), int(players[1]))
    with pytest.raises(ValueError):                                       # more pieces than the variant has
        env.reset(torch.ones((n, 10, 10), dtype=torch.int8), torch.ones((n, 10, 10), dtype=torch.int8))
    env.close(); penv.close(); packed.close()


@pytest.mark.parametrize('name', ['barrage', 'tiny', 'fives'])
def test_packed_states_expand_equals_get_next_state(name):
    """Search on packed records: pool-to-pool expansion (sgx_expand) with a parent index gives exactly the successors the
    int64 get_next_state gives (which the other tests pin to the oracle); invalid actions leave a copy of the parent."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    from stratego_env_amd.vec_env import VecStrategoEnv
    n_par, n = 48, 192
    env = VecStrategoEnv(name, n_par, seed=21, auto_reset=True)
    env.reset()
    env.rollout_steps(9)
    states, players = env.export_state()
    ppar = BatchedStrategoProceduralEnv(name, n_par)
    parents = ppar.pack(states, players)
    assert int(parents.sanitised.sum()) == 0
    rs = np.random.RandomState(3)
    pidx = torch.from_numpy(rs.randint(0, n_par, size=n).astype(np.int32)).cuda()
    masks = ppar.get_valid_moves_as_1d_mask(states, players)[pidx.long()]
    # a random valid action for most children, garbage for some
    pick = torch.multinomial((masks != 0).float(), 1).squeeze(1).to(torch.int32)
    garbage = torch.from_numpy(rs.rand(n) < 0.2).cuda()
    acts = torch.where(garbage, torch.randint(0, masks.shape[1], (n,), device='cuda', dtype=torch.int32), pick)
    pch = BatchedStrategoProceduralEnv(name, n)
    children = pch.new_packed()
    mask_out = torch.empty((n, masks.shape[1]), dtype=torch.uint8, device='cuda')
    valid, child_players = children.expand(parents, acts, parent_index=pidx, mask_1d_out=mask_out)
    want_states, want_players, want_valid = pch.get_next_state(states[pidx.long()], players[pidx.long()], acts)
    got_states, got_players = children.unpack()
    assert torch.equal(valid, want_valid) and bool((~valid).any()) and bool(valid.any())
    assert torch.equal(got_states, want_states) and torch.equal(got_players, want_players) and torch.equal(child_players, want_players)
    assert torch.equal(mask_out, pch.get_valid_moves_as_1d_mask(want_states, want_players))
    assert torch.equal(children.valid_moves_as_1d_mask(), mask_out)
    # the same expansion WITHOUT a mask: the launch only asks whether the next mover has a move (gen_mask<..., ANY>) -- same successors
    bare = pch.new_packed()
    valid2, players2 = bare.expand(parents, acts, parent_index=pidx)
    bs, bp = bare.unpack()
    assert torch.equal(valid2, want_valid) and torch.equal(bs, want_states) and torch.equal(bp, want_players) and torch.equal(players2, want_players)
    bare.close()
    # the parents are untouched; copy_from scatters records between pools
    ps, pp = parents.unpack()
    assert torch.equal(ps, states) and torch.equal(pp, players)
    pool = pch.new_packed()
    dst = torch.arange(n - 1, -1, -1, dtype=torch.int32, device='cuda')
    pool.copy_from(children, dst_index=dst)
    cs, cp = pool.unpack()
    assert torch.equal(cs.flip(0), got_states) and torch.equal(cp.flip(0), got_players)
    for x in (env, ppar, pch, parents, children, pool):
        x.close()


def test_stuck_opponent_behind_a_vetoed_cell_without_a_mask():
    """The opponent-stuck ending (impl:1031-1036) when the next mover's only way out is the two-square-vetoed cell (impl:439-445): a
    sergeant is stuck (game over), a scout walks on past the vetoed cell (game goes on) unless the cell behind it is blocked too.
    Through every path that answers 'does a move exist' without building the mask (sgx_expand, logic-only sgx_step) and through the
    ones that build it, against the oracle."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS['barrage']
    ru = orc.OracleRules(10, 10)
    states, expect_over = [], []
    for piece, behind in ((4, 0), (2, 0), (2, 12), (2, -1), (4, -1)):     # X, what stands on (7,9): empty / own bomb / an ENEMY piece there
        st = np.zeros((34, 10, 10), dtype=np.int64)
        for r, c in v.obstacle_locations:
            st[2, r, c] = 1
        st[5, 0, 0], st[5, 1, 0] = 10, v.max_turns
        st[0, 0, 0], st[0, 0, 9] = 4, 11                                   # player +1: a sergeant (moves now) and the flag
        st[1, 9, 9], st[1, 9, 8], st[1, 9, 7] = piece, 12, 11              # player -1: X in the corner, a bomb beside it, the flag
        if behind > 0:
            st[1, 7, 9] = behind
        elif behind < 0:
            st[0, 7, 9] = 5                                                 # an enemy (player +1) lieutenant: X can attack it ... only a scout reaches it
        for layer in (0, 1):
            st[3 + layer] = np.where(st[layer] != 0, 13, 0)
            st[32 + layer] = (st[layer] != 0)
        st[7, 9, 9], st[7, 8, 9] = -3, 1                                    # X came (8,9) -> (9,9) and may not go straight back
        states.append(st)
    states = np.stack(states)
    n = len(states)
    players = np.ones(n, dtype=np.int8)
    acts = np.full(n, ru.get_action_1d_index_from_positions(0, 0, 1, 0), dtype=np.int64)
    want = [ru.get_next_state(states[i], 1, int(acts[i])) for i in range(n)]
    over = [int(w[0][5, 0, 1]) for w in want]
    assert over == [1, 0, 1, 0, 1], over                                    # sergeant stuck; scout passes; scout blocked behind; scout attacks; sergeant stuck
    pe = BatchedStrategoProceduralEnv('barrage', n)
    ns, npl, ok = pe.get_next_state(states, players, acts)
    parents = pe.pack(states, players)
    assert int(parents.sanitised.sum()) == 0
    kids = pe.new_packed()
    valid, kp = kids.expand(parents, acts)                                   # no mask, no next action: the any-move question
    ks, kpl = kids.unpack()
    vec = VecStrategoEnv('barrage', n, auto_reset=False, human_inits=False)
    vec.import_state(states, players)
    a32 = torch.from_numpy(acts.astype(np.int32)).to(vec.device)
    from stratego_env_amd import _lib
    vec.step(a32, emit_obs=False, emit_mask=False, flags=_lib.STEP_ACTIONS_1D)   # logic-only step: the same question
    vs, vpl = vec.export_state()
    for i in range(n):
        for got, gpl, what in ((ns, npl, 'get_next_state'), (ks, kpl, 'expand'), (vs, vpl, 'logic-only step')):
            assert np.array_equal(got[i].cpu().numpy(), want[i][0]) and int(gpl[i]) == want[i][1], (i, what)
        assert bool(ok[i]) and bool(valid[i])
    assert vec.done.cpu().numpy().tolist() == over
    for x in (pe, parents, kids, vec):
        x.close()


def test_directed_combat_matrix_and_quirks_vs_oracle():
    """SURVEY A.8 quirks on constructed positions (tests.helpers.directed_positions): next state, next mover, validity, game
    result and both observations against the oracle (which tools/oracle/check_directed_vs_reference.py pins to the reference
    on the same positions)."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    from tests.helpers import directed_positions
    ru = orc.OracleRules(4, 4)
    states, players, actions = directed_positions()
    n = len(states)
    penv = BatchedStrategoProceduralEnv('tiny', n)
    ns, npl, ok = penv.get_next_state(states, players, actions)
    ns, npl, ok = ns.cpu().numpy(), npl.cpu().numpy(), ok.cpu().numpy()
    ended = penv.get_game_ended(torch.from_numpy(ns), torch.from_numpy(npl)).cpu().numpy()
    po = penv.get_partially_observable_observation_extended_channels(torch.from_numpy(ns), torch.from_numpy(npl)).cpu().numpy()
    fo = penv.get_fully_observable_observation_extended_channels(torch.from_numpy(ns), torch.from_numpy(npl)).cpu().numpy()
    wins = invalid_endings = 0
    for i in range(n):
        want, wpl = ru.get_next_state(states[i], int(players[i]), int(actions[i]))       # every constructed move is legal
        assert ok[i], i
        assert np.array_equal(ns[i], want), (i, int(players[i]), int(actions[i]))
        assert npl[i] == wpl
        assert np.float32(ended[i]) == np.float32(ru.get_game_ended(want, wpl)), i
        assert po[i].tobytes() == ru.get_partially_observable_observation_extended_channels(want, wpl).tobytes(), i
        assert fo[i].tobytes() == ru.get_fully_observable_observation_extended_channels(want, wpl).tobytes(), i
        wins += int(want[5, 0, 1] == 1 and want[5, 0, 2] != 0)
        invalid_endings += int(want[5, 1, 1] == 1)
    assert wins > 0 and invalid_endings > 0
    penv.close()


def test_heuristic_rewards_match_reference_golden():
    """get_heuristic_rewards_from_move vs values recorded from the reference's _get_heuristic_rewards_from_move (impl:852-891)."""
    import json
    import os
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    from tests.helpers import GOLDEN, directed_positions
    with open(os.path.join(GOLDEN, 'heuristic_rewards.json')) as f:
        g = json.load(f)
    states, players, actions = directed_positions()
    rm = np.random.RandomState(g['matrix_seed']).rand(14, 14).astype(np.float32)
    penv = BatchedStrategoProceduralEnv('tiny', len(states))
    got = penv.get_heuristic_rewards_from_move(states, players, actions, rm).cpu().numpy()
    assert np.array_equal(got, np.asarray(g['rewards'], dtype=np.float32))
    penv.close()


@pytest.mark.parametrize('name', ['barrage', 'medium', 'octa_barrage', 'standard2', 'fives', 'tiny', 'micro'])
def test_general_states_are_reproduced_not_sanitised(name):
    """penv's pure functions accept ANY int64 [34,R,C] (penv:74-155).  States the packed record cannot carry are redone by
    sgx_step_states' second pass on the general-state variant of the kernels: next state, validity, both mask encodings and the raw
    partial observation equal OracleRules', and no state comes back flagged."""
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    v = VARIANTS[name]
    rs = np.random.RandomState(23)
    n = 64 if name != 'standard2' else 24
    states, players = general_states(name, n, rs)
    ru = orc.OracleRules(v.rows, v.columns)
    pe = BatchedStrategoProceduralEnv(name, n)
    # the first pass alone would have flagged every one of them
    san = __import__('torch').zeros(n, dtype=__import__('torch').uint8, device=pe.device)
    pe._vec.import_state_checked(states, players, san)
    assert int(san.sum()) == n
    for pl_all in (players, -players):
        m1 = pe.get_valid_moves_as_1d_mask(states, pl_all).cpu().numpy()
        assert int(pe.last_sanitised.sum()) == 0
        ms = pe.get_valid_moves_as_spatial_mask(states, pl_all).cpu().numpy()
        assert int(pe.last_sanitised.sum()) == 0
        po = pe.get_partially_observable_observation_extended_channels(states, pl_all).cpu().numpy()
        assert int(pe.last_sanitised.sum()) == 0
        # the other observation kinds (79-channel, 'original' channels): three launches on packed records first, then the same second pass
        others = []
        for fn_name in ('get_fully_observable_observation_extended_channels', 'get_partially_observable_observation',
                        'get_fully_observable_observation'):
            others.append((fn_name, getattr(pe, fn_name)(states, pl_all).cpu().numpy()))
            assert int(pe.last_sanitised.sum()) == 0, (name, fn_name)
        for e in range(n):
            p = int(pl_all[e])
            for fn_name, got in others:
                assert got[e].tobytes() == getattr(ru, fn_name)(states[e], p).tobytes(), (name, e, fn_name)
            assert np.array_equal(m1[e], ru.get_valid_moves_as_1d_mask(states[e], p)), (name, e, '1d mask')
            assert np.array_equal(ms[e], ru.get_valid_moves_as_spatial_mask(states[e], p)), (name, e, 'spatial mask')
            assert po[e].tobytes() == ru.get_partially_observable_observation_extended_channels(states[e], p).tobytes(), (name, e, 'obs')
    n_valid = 0
    for rnd in range(5):
        acts = np.zeros(n, dtype=np.int64)
        for e in range(n):
            mask = ru.get_valid_moves_as_1d_mask(states[e], int(players[e]))
            acts[e] = rs.choice(np.flatnonzero(mask)) if rs.rand() < 0.7 else rs.randint(-3, ru.action_size + 3)
        for osc in (False, True):
            ns, npl, valid = pe.get_next_state(states, players, acts, allow_piece_oscillation=osc)
            assert int(pe.last_sanitised.sum()) == 0
            ns, npl, valid = ns.cpu().numpy(), npl.cpu().numpy(), valid.cpu().numpy()
            v2 = pe.is_move_valid_by_1d_index(states, players, acts, allow_piece_oscillation=osc).cpu().numpy()
            for e in range(n):
                want_valid = ru.is_move_valid_by_1d_index(states[e], int(players[e]), int(acts[e]), allow_piece_oscillation=osc)
                assert bool(valid[e]) == want_valid == bool(v2[e]), (name, rnd, e, acts[e], osc)
                if want_valid:
                    n_valid += 1
                    w, wp = ru.get_next_state(states[e], int(players[e]), int(acts[e]), allow_piece_oscillation=osc)
                    assert np.array_equal(ns[e], w) and npl[e] == wp, (name, rnd, e, 'next state', np.argwhere(ns[e] != w)[:4])
                else:
                    assert np.array_equal(ns[e], states[e]) and npl[e] == players[e]
    assert n_valid > n
    pos = rs.randint(-1, max(v.rows, v.columns) + 1, size=(n, 4))
    got = pe.is_move_valid_by_position(states, players, pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3]).cpu().numpy()
    for e in range(n):
        assert bool(got[e]) == ru.is_move_valid_by_position(states[e], int(players[e]), *[int(x) for x in pos[e]])
    # values outside a layer's range stay flagged (and only those)
    bad = states.copy()
    bad[0, 0, 0, 0] = 14
    bad[1, 9, 1, 1] = -1
    pe.get_valid_moves_as_1d_mask(bad, players)
    flagged = pe.last_sanitised.cpu().numpy()
    assert flagged[0] == 1 and flagged[1] == 1 and int(flagged.sum()) == 2
    # switched off, every such state is flagged again
    pe._vec._L.sgx_set_general_states(pe._vec._h, 0)
    pe.get_valid_moves_as_1d_mask(states, players)
    assert int(pe.last_sanitised.sum()) == n
    pe.close()


def test_step_states_refuses_outputs_that_alias_its_inputs_while_the_general_pass_is_on():
    """The general-state pass reads the caller's input again after the first pass has written the outputs (round-4 advisor finding):
    stepping in place would redo flagged states from their own successors.  SGX_EINVAL names the way out; with the pass off, in place works
    and equals the out-of-place result."""
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    name, n = 'barrage', 32
    v = VARIANTS[name]
    pe = BatchedStrategoProceduralEnv(name, n)
    cv = oracle_cvariant(name)
    states = np.stack([orc.reset_state(cv, 5, e, 0) for e in range(n)])
    players = np.ones(n, dtype=np.int8)
    ru = orc.OracleRules(v.rows, v.columns)
    acts = np.asarray([np.flatnonzero(ru.get_valid_moves_as_1d_mask(states[e], 1))[e % 3] for e in range(n)], dtype=np.int64)
    want, want_pl, ok = pe.get_next_state(states, players, acts)
    assert bool(ok.all())
    st = torch.from_numpy(states).to(pe.device)
    pl = torch.from_numpy(players).to(pe.device)
    a = torch.from_numpy(acts.astype(np.int32)).to(pe.device)
    vec = pe._vec
    io = vec._fill_io(a, False, False, False, _lib.STEP_ACTIONS_1D)
    io.auto_reset = 0
    args = lambda so, po: (vec._h, st.data_ptr(), pl.data_ptr(), pe.last_sanitised.data_ptr(), io, so.data_ptr(), po.data_ptr(), 2, vec._stream())
    other_pl = torch.empty_like(pl)
    assert vec._L.sgx_step_states(*args(st, other_pl)) != 0
    assert b'overlap the inputs' in vec._L.sgx_last_error()
    assert vec._L.sgx_step_states(*args(torch.empty_like(st), pl)) != 0                       # the players alone overlap
    half = st.view(-1)[st.numel() // 2:]                                                      # a partial overlap is an overlap
    assert vec._L.sgx_step_states(vec._h, st.data_ptr(), pl.data_ptr(), pe.last_sanitised.data_ptr(), io, half.data_ptr(), other_pl.data_ptr(), 2, vec._stream()) != 0
    assert torch.equal(st.cpu(), torch.from_numpy(states))                                    # nothing was launched
    _lib.check(vec._L.sgx_set_general_states(vec._h, 0), vec._L)
    _lib.check(vec._L.sgx_step_states(*args(st, pl)), vec._L)                                 # in place, pass off: fine
    assert torch.equal(st, want) and torch.equal(pl, want_pl)
    pe.close()


def test_loaded_scope_gives_the_same_results_for_general_states():
    """`with env.loaded(states, players)` is an optimisation, not another semantics (round-4 advisor finding): when the import had to alter
    any of the held states nothing is reused, every call inside the scope goes through the general-state pass like a call outside, and
    last_sanitised / strict mean the same on both paths."""
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    name, n = 'barrage', 48
    v = VARIANTS[name]
    rs = np.random.RandomState(77)
    states, players = general_states(name, n, rs)
    ru = orc.OracleRules(v.rows, v.columns)
    pe = BatchedStrategoProceduralEnv(name, n)
    pe.strict = True                                              # must not raise: after the general pass nothing is altered
    acts = np.asarray([rs.choice(np.flatnonzero(ru.get_valid_moves_as_1d_mask(states[e], int(players[e])))) for e in range(n)], dtype=np.int64)
    out_ns, out_pl, out_ok = pe.get_next_state(states, players, acts)
    out_mask = pe.get_valid_moves_as_1d_mask(states, players)
    out_obs = pe.get_partially_observable_observation_extended_channels(states, players)
    with pe.loaded(states, players):
        assert not pe._in_loaded_scope(states, players)           # general states: no reuse inside the scope
        in_mask = pe.get_valid_moves_as_1d_mask(states, players)
        assert int(pe.last_sanitised.sum()) == 0
        in_obs = pe.get_partially_observable_observation_extended_channels(states, players)
        in_ns, in_pl, in_ok = pe.get_next_state(states, players, acts)
        assert int(pe.last_sanitised.sum()) == 0
    import torch
    assert torch.equal(in_mask, out_mask) and torch.equal(in_obs, out_obs) and torch.equal(in_ns, out_ns) and torch.equal(in_pl, out_pl)
    assert bool(in_ok.all()) and bool(out_ok.all())
    for e in range(0, n, 5):
        w, wp = ru.get_next_state(states[e], int(players[e]), int(acts[e]))
        assert np.array_equal(in_ns[e].cpu().numpy(), w) and int(in_pl[e]) == wp
    # reachable states: the scope does reuse its import, with identical results
    cv = oracle_cvariant(name)
    reach = np.stack([orc.reset_state(cv, 9, e, 0) for e in range(n)])
    ones = np.ones(n, dtype=np.int8)
    m_out = pe.get_valid_moves_as_1d_mask(reach, ones)
    with pe.loaded(reach, ones):
        assert pe._in_loaded_scope(reach, ones)
        assert torch.equal(pe.get_valid_moves_as_1d_mask(reach, ones), m_out)
    pe.close()
