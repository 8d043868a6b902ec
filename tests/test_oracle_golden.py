"""The oracle (oracle/stratego_oracle.c) replayed against vectors generated from the reference.

This is what pins the oracle on any box (the GPU box has no /root/reference); in the build container
tools/oracle/check_oracle_vs_reference.py additionally diffs it against the live reference.
"""
import hashlib

import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd import setups as S
from stratego_env_amd.config import VARIANTS
from tests.helpers import (digest_obs, load_expanded, load_games, load_kat, load_variants_json, oracle_cvariant,
                           oracle_env)

GAME_SETS = ['barrage', 'standard', 'short_barrage', 'short_standard', 'octa_barrage', 'medium', 'fives', 'tiny',
             'micro', 'standard2']


def test_norm_constants_match_reference():
    ref = load_variants_json()['variants']
    for name, r in ref.items():
        mids, ranges = orc.p_obs_norm_constants(VARIANTS[name].piece_counts)
        assert np.array_equal(mids, np.asarray(r['p_obs_mids'], dtype=np.float32)), name
        assert np.array_equal(ranges, np.asarray(r['p_obs_ranges'], dtype=np.float32)), name


def test_index_tables():
    t = np.load('tests/golden/index_tables.npz')
    for name in ('barrage', 'octa_barrage', 'medium', 'fives', 'tiny', 'micro'):
        v = VARIANTS[name]
        ru = orc.OracleRules(v.rows, v.columns)
        s2o = t[name + '_spatial_to_1d']
        K = v.spatial_channels
        got = np.asarray([ru.get_action_1d_index_from_spatial_index((a // K // v.columns, a // K % v.columns, a % K))
                          for a in range(v.num_spatial_actions)], dtype=np.int32)
        assert np.array_equal(got, s2o), name
        flip = t[name + '_flip_1d']
        got = np.asarray([ru.get_action_1d_index_from_player_perspective(i, -1) for i in range(v.action_size)], dtype=np.int32)
        assert np.array_equal(got, flip), name


@pytest.mark.parametrize('name', GAME_SETS)
def test_replay_golden_games(name):
    g = load_games(name)
    off = g['offsets']
    n_games = len(off) - 1
    # keep the CPU suite short: all toy games, a slice of the long ones
    limit = n_games
    env = oracle_env(name)
    for gi in range(min(n_games, limit)):
        obs = env.reset(g['p1_maps'][gi].astype(np.int64), g['p2_maps'][gi].astype(np.int64))
        assert digest_obs(obs) == int(g['init_digests'][gi]), (name, gi, 'reset')
        for k in range(off[gi], off[gi + 1]):
            p = list(obs.keys())[0] if len(obs) == 1 else env.player
            a = int(g['actions'][k])
            if g['errors'][k]:
                with pytest.raises(ValueError):
                    env.step({env.player: a})
                continue
            obs, rew, done, info = env.step({env.player: a})
            assert digest_obs(obs) == int(g['digests'][k]), (name, gi, k)
            assert bool(done['__all__']) == bool(g['dones'][k])
            assert env.player == int(g['players'][k])
            assert (float(rew.get(1, 0.0)), float(rew.get(-1, 0.0))) == tuple(float(x) for x in g['rewards'][k])
        assert np.array_equal(env.state, g['final_states'][gi].astype(np.int64)), (name, gi, 'final state')
        if g['finished'][gi]:
            assert info[1]['game_result_was_invalid'] == bool(g['ending_invalid'][gi])


@pytest.mark.parametrize('name', ['barrage', 'octa_barrage', 'medium', 'fives', 'tiny', 'micro'])
def test_replay_expanded_games(name):
    ex = load_expanded(name)
    n = len([k for k in ex if k.endswith('_actions')])
    env = oracle_env(name)
    for gi in range(n):
        pre = 'g%d_' % gi
        masks, obss, slot_player = ex[pre + 'masks'], ex[pre + 'obs'], ex[pre + 'slot_player']
        obs = env.reset(ex[pre + 'p1_map'].astype(np.int64), ex[pre + 'p2_map'].astype(np.int64))
        slot = 0
        assert np.array_equal(obs[1]['valid_actions_mask'], masks[slot]) and obs[1]['partial_observation'].tobytes() == obss[slot].tobytes()
        slot += 1
        for k, a in enumerate(ex[pre + 'actions']):
            obs, rew, done, info = env.step({env.player: int(a)})
            for pl in sorted(obs.keys(), reverse=True):
                assert slot_player[slot] == pl
                assert np.array_equal(obs[pl]['valid_actions_mask'], masks[slot]), (name, gi, k)
                assert obs[pl]['partial_observation'].tobytes() == obss[slot].tobytes(), (name, gi, k)
                slot += 1
            assert bool(done['__all__']) == bool(ex[pre + 'dones'][k])
        assert slot == len(masks)
        assert np.array_equal(env.state, ex[pre + 'final_state'].astype(np.int64))


@pytest.mark.parametrize('name', ['barrage', 'standard'])
def test_survey_known_answers(name):
    """rolling sha256 of a full game under the (n*7919 mod nvalid) policy."""
    kat = load_kat()[name]
    v = VARIANTS[name]
    m1, m2 = S.own_side_maps(S.codes_from_string(kat['setup1']), S.codes_from_string(kat['setup2']), v.rows, v.columns,
                             v.initial_state_usable_rows)
    env = oracle_env(name)
    obs = env.reset(m1, m2)
    m0 = obs[1]['valid_actions_mask']
    assert [int(x) for x in np.flatnonzero(m0)] == kat['init_valid']
    assert hashlib.sha256(obs[1]['partial_observation'].tobytes()).hexdigest()[:16] == kat['init_obs_sha']
    assert hashlib.sha256(m0.astype(np.uint8).tobytes()).hexdigest()[:16] == kat['init_mask_sha']
    h = hashlib.sha256()
    n = 0
    while True:
        p = list(obs.keys())[0]
        m = obs[p]['valid_actions_mask']
        a = int(np.flatnonzero(m)[(n * 7919) % int(m.sum())])
        obs, rew, done, info = env.step({p: a})
        n += 1
        for pl in sorted(obs.keys()):
            h.update(obs[pl]['valid_actions_mask'].astype(np.uint8).tobytes())
            h.update(obs[pl]['partial_observation'].tobytes())
        if done['__all__']:
            break
    assert n == kat['steps'] and h.hexdigest()[:16] == kat['rolling_sha']
    assert [rew[1], rew[-1]] == kat['rewards'] and info[1]['game_result_was_invalid'] == kat['invalid']


def test_two_square_rule_known_answer():
    kat = load_kat()['two_square']
    ru = orc.OracleRules(4, 4)
    m1 = np.zeros((4, 4), dtype=np.int64); m2 = np.zeros((4, 4), dtype=np.int64)
    m1[0, 0] = 5; m1[0, 3] = 11; m2[0, 0] = 5; m2[0, 3] = 11
    st = ru.create_initial_state(np.zeros((4, 4), dtype=np.int64), m1, m2, 100)
    pl = 1
    for (s, e) in [((0, 0), (1, 0)), ((3, 3), (2, 3)), ((1, 0), (0, 0)), ((2, 3), (3, 3)), ((0, 0), (1, 0)), ((3, 3), (2, 3))]:
        st, pl = ru.get_next_state(st, pl, ru.get_action_1d_index_from_positions(*s, *e))
    assert st[6].tolist() == kat['p1_recent']
    assert ru.is_move_valid_by_position(st, 1, 1, 0, 0, 0) == kat['fourth_oscillation_valid']
    assert [int(x) for x in np.flatnonzero(ru.get_valid_moves_as_spatial_mask(st, 1))] == kat['p1_mask_after']


def test_rollout_rule_reproduces_golden_barrage_games():
    """The bench's synthetic-rollout rule (RNG-chosen setups + k-th valid action) on the oracle reproduces
    the held-out golden games, whose outputs were recorded from the reference."""
    g = load_games('barrage')
    table = S.load_setup_table('barrage')
    cv = oracle_cvariant('barrage', setups=table)
    env = oracle_env('barrage')
    off = g['offsets']
    for gi in range(12):
        seed = int(g['seeds'][gi])
        m1, m2 = orc.sample_setup(cv, seed, 0, 0)
        assert np.array_equal(m1, g['p1_maps'][gi]) and np.array_equal(m2, g['p2_maps'][gi])
        obs = env.reset(m1, m2)
        for k in range(off[gi], off[gi + 1]):
            p = env.player
            a = orc.sample_action(obs[p]['valid_actions_mask'].astype(np.uint8), seed, 0, 0, int(env.state[5, 0, 0]))
            assert a == int(g['actions'][k])
            obs, rew, done, info = env.step({p: a})
            assert digest_obs(obs) == int(g['digests'][k])
        assert done['__all__']
