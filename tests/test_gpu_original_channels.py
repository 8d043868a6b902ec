"""obs_channel_mode='original' (the deprecated 32 / 33 value-channel observations, maenv:67, 368-375) on the GPU: vs the
oracle, vs vectors recorded from the reference (tests/golden/games_orig_*.npz), and through the procedural operator API."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd import GameVersions, ObservationModes
from stratego_env_amd.config import VARIANTS
from tests.helpers import GOLDEN, _load_npz, oracle_cvariant
from tests.test_gpu_full_obs import check_both_obs_vs_oracle
from tests.test_gpu_parity import _table
from tests.test_oracle_golden_both import digest_both

pytestmark = pytest.mark.gpu
MASK, POBS, FOBS = 'valid_actions_mask', 'partial_observation', 'full_observation'


@pytest.mark.parametrize('name,n_envs,n_steps', [('barrage', 32, 400), ('standard', 6, 200), ('tiny', 48, 150), ('micro', 48, 80),
                                                 ('fives', 32, 150), ('octa_barrage', 16, 200), ('medium', 8, 150),
                                                 ('standard2', 2, 60)])
def test_original_both_obs_bit_exact_vs_oracle(name, n_envs, n_steps):
    check_both_obs_vs_oracle(name, n_envs, n_steps, 'original')


@pytest.mark.parametrize('name,n_envs,n_steps', [('barrage', 64, 300), ('micro', 64, 60), ('fives', 16, 100)])
def test_original_partial_only_bit_exact_vs_oracle(name, n_envs, n_steps):
    """PARTIALLY_OBSERVABLE + original channels (no full observation rendered)."""
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[name]
    seed, g0 = 0x0816 + len(name), 9000
    env = VecStrategoEnv(name, n_envs, seed=seed, env_id_offset=g0, auto_reset=True, obs_channel_mode='original')
    assert tuple(env.obs.shape) == (n_envs, v.rows, v.columns, 32) and env.fobs is None
    cv = oracle_cvariant(name, setups=_table(name))
    oenvs = []
    for e in range(n_envs):
        oe = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts, obs_channel_mode='original')
        oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, 0))
        oe.game_no = 0
        oenvs.append(oe)
    env.reset()
    obs_h = env.obs.cpu().numpy()
    for e, oe in enumerate(oenvs):
        assert oe._obs(1)[POBS].tobytes() == obs_h[e].tobytes()
    env.sample_valid_actions()
    for t in range(n_steps):
        acts = env.next_actions.cpu().numpy().copy()
        env.rollout_step()
        obs_h, mask_h, done_h = env.obs.cpu().numpy(), env.mask.cpu().numpy(), env.done.cpu().numpy()
        for e, oe in enumerate(oenvs):
            o, rew, done, info = oe.step({oe.player: int(acts[e])})
            assert bool(done_h[e]) == done['__all__']
            if done['__all__']:
                oe.game_no += 1
                o = oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, oe.game_no))
            p = oe.player
            assert o[p][POBS].tobytes() == obs_h[e].tobytes(), (name, t, e)
            assert np.array_equal(o[p][MASK], mask_h[e])
    env.close()


@pytest.mark.parametrize('name', ['barrage', 'tiny', 'micro', 'fives', 'octa_barrage'])
def test_facade_original_mode_replays_reference_goldens(name):
    from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv
    from tests.test_gpu_facade import _state_from_maps
    g = _load_npz(os.path.join(GOLDEN, 'games_orig_%s.npz' % name))
    env = StrategoMultiAgentEnv({'version': GameVersions(name), 'obs_channel_mode': 'original'})    # BOTH is the default mode
    assert env.observation_space.spaces[POBS].shape == (env.rows, env.columns, 32)
    assert env.observation_space.spaces[FOBS].shape == (env.rows, env.columns, 33)
    off = g['offsets']
    for gi in range(min(4, len(off) - 1)):
        obs = env.reset(initial_state_override=_state_from_maps(name, g['p1_maps'][gi], g['p2_maps'][gi]))
        assert obs[1][POBS].shape == (env.rows, env.columns, 32) and obs[1][FOBS].shape == (env.rows, env.columns, 33)
        assert digest_both(obs) == int(g['init_digests'][gi])
        # the reference's public (de)normalisation helpers (maenv:499-511) round-trip to integer channel values
        raw = env.denormalize_p_observation(obs[1][POBS])
        assert np.allclose(raw, np.round(raw), atol=1e-5)
        assert np.allclose(env.normalize_p_observation(np.round(raw).astype(np.float32)), obs[1][POBS], atol=1e-6)
        for k in range(off[gi], off[gi + 1]):
            obs, rew, done, info = env.step({env.player: int(g['actions'][k])})
            assert digest_both(obs) == int(g['digests'][k]), (name, gi, k)
            assert done['__all__'] == bool(g['dones'][k])
    env.close()
    with pytest.raises(ValueError):
        StrategoMultiAgentEnv({'version': GameVersions(name), 'obs_channel_mode': 'compact'})


@pytest.mark.parametrize('name', ['barrage', 'micro', 'fives'])
def test_procedural_original_observations_and_heuristic_rewards(name):
    """get_{partially,fully}_observable_observation (penv:157-164) raw, and _get_heuristic_rewards_from_move (impl:852-891)."""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    v = VARIANTS[name]
    N = 24
    cv = oracle_cvariant(name, setups=_table(name))
    ru = orc.OracleRules(v.rows, v.columns)
    rng = np.random.RandomState(5)
    states, players = [], []
    for e in range(N):                              # states from random oracle play, various depths
        st, pl = orc.reset_state(cv, 77, e, 0), 1
        for t in range(int(rng.randint(0, 40))):
            m = ru.get_valid_moves_as_1d_mask(st, pl)
            if ru.get_game_ended(st, pl) != 0 or m[-1]:
                break
            a = int(rng.choice(np.flatnonzero(m)))
            st, pl = ru.get_next_state(st, pl, a)
        states.append(st)
        players.append(pl)
    states, players = np.stack(states), np.asarray(players, dtype=np.int8)
    penv = BatchedStrategoProceduralEnv(name, N)
    po = penv.get_partially_observable_observation(states, players).cpu().numpy()
    fo = penv.get_fully_observable_observation(states, players).cpu().numpy()
    assert po.shape == (N, v.rows, v.columns, 32) and fo.shape == (N, v.rows, v.columns, 33)
    for e in range(N):
        assert po[e].tobytes() == ru.get_partially_observable_observation(states[e], int(players[e])).tobytes(), e
        assert fo[e].tobytes() == ru.get_fully_observable_observation(states[e], int(players[e])).tobytes(), e
    # heuristic rewards: reward_matrix[moved type, destination enemy type]; no-op -> 0
    rm = rng.rand(13, 13).astype(np.float32)
    acts, want = [], []
    for e in range(N):
        m = ru.get_valid_moves_as_1d_mask(states[e], int(players[e]))
        a = int(rng.choice(np.flatnonzero(m)))
        acts.append(a)
        if a == ru.action_size - 1:
            want.append(0.0)
        else:
            sr, sc, er, ec = ru.get_action_positions_from_1d_index(a)
            own, enemy = (0, 1) if players[e] == 1 else (1, 0)
            want.append(float(rm[states[e][own, sr, sc], states[e][enemy, er, ec]]))
    got = penv.get_heuristic_rewards_from_move(states, players, np.asarray(acts), rm).cpu().numpy()
    assert np.array_equal(got, np.asarray(want, dtype=np.float32))


def test_operator_debugging_aids_match_reference_goldens(capsys):
    """print_board_to_console, get_dict_of_valid_moves_by_position, the serializable strings and the
    player_perspective flag of get_valid_moves_as_1d_mask (penv:74-85, 175-214) vs outputs recorded from the reference
    (tools/oracle/gen_golden_curriculum.py -> tests/golden/board_utils.json)."""
    import hashlib
    import json
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    with open(os.path.join(GOLDEN, 'board_utils.json')) as f:
        recs = json.load(f)
    cur = _load_npz(os.path.join(GOLDEN, 'curriculum_barrage.npz'))
    states = np.stack([cur['state'][r['index']] for r in recs])
    n = len(recs)
    penv = BatchedStrategoProceduralEnv('barrage', n)
    fo = penv.get_serializable_string_for_fully_observable_state(states)
    po = penv.get_serializable_string_for_partially_observable_state(states)
    for pl in (1, -1):
        players = np.full(n, pl, dtype=np.int8)
        dicts = penv.get_dict_of_valid_moves_by_position(states, players)
        pp = penv.get_valid_moves_as_1d_mask(states, players, player_perspective=True).cpu().numpy().astype(np.uint8)
        for i, r in enumerate(recs):
            assert dicts[i] == r['moves_dict_%d' % pl], (i, pl)
            assert hashlib.sha256(np.ascontiguousarray(pp[i]).tobytes()).hexdigest()[:16] == r['mask1d_pp_%d' % pl], (i, pl)
    for i, r in enumerate(recs):
        assert hashlib.sha256(fo[i]).hexdigest()[:16] == r['fo_string_sha']
        assert hashlib.sha256(po[i]).hexdigest()[:16] == r['po_string_sha']
        for po_flag, hide in ((False, True), (True, False)):
            capsys.readouterr()
            penv.print_board_to_console(states[i], partially_observable=po_flag, hide_still_piece_markers=hide)
            assert capsys.readouterr().out == r['print_po%d_hide%d' % (po_flag, hide)]


@pytest.mark.parametrize('table', ['curriculum_barrage.npz', 'curriculum_barrage_contiguous.h5', 'curriculum_barrage_chunked_gzip.h5',
                                   'curriculum_barrage_latest.h5'])
def test_facade_curriculum_start_states_replay_reference_goldens(table):
    """curriculum_start_states_path (maenv:341-346, 519-527; util.py:372-387): same np.random consumption, start state,
    first mover, player relabelling and per-step outputs as recorded from the reference."""
    import hashlib
    import json
    import random
    from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

    def obs_digest(obs):
        h = hashlib.sha256()
        for p in sorted(obs.keys()):
            h.update(np.asarray(obs[p][MASK]).astype(np.uint8).tobytes())
            h.update(np.ascontiguousarray(obs[p][POBS], dtype=np.float32).tobytes())
        return h.hexdigest()[:16]

    with open(os.path.join(GOLDEN, 'curriculum.json')) as f:
        cases = json.load(f)
    path = os.path.join(GOLDEN, table)       # (.h5: real HDF5 files written by h5py, read by the package's own reader here: no h5py in the image)
    for case in cases:
        np.random.seed(case['seed'])
        random.seed(case['seed'])
        env = StrategoMultiAgentEnv({'version': GameVersions.BARRAGE, 'observation_mode': ObservationModes.PARTIALLY_OBSERVABLE,
                                     'curriculum_start_states_path': path,
                                     'same_start_pos_everytime': case['same_start_pos_everytime']})
        assert env.use_curriculum_inits and env.random_player_assignment
        for g in case['games']:
            obs = env.reset()
            assert int(list(obs.keys())[0]) == g['first_key'] and env.player == g['player']
            assert sha(env.state) == g['state'] and obs_digest(obs) == g['init']
            for t, srec in enumerate(g['steps']):
                k = list(obs.keys())[0]
                valid = np.flatnonzero(obs[k][MASK].reshape(-1))
                a = int(valid[(t * 7919) % len(valid)])
                assert a == srec['action']
                obs, rew, done, info = env.step({k: a})
                assert sorted(int(x) for x in obs.keys()) == srec['keys'] and obs_digest(obs) == srec['digest']
                assert bool(done['__all__']) == srec['done']
                assert {str(kk): float(vv) for kk, vv in rew.items()} == srec['rewards']
        env.close()
