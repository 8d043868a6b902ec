"""Fully-observable observation (79 channels) and BOTH_OBSERVATIONS mode on the GPU: vs the oracle and vs vectors
recorded from the reference (tests/golden/games_both_*.npz)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd import GameVersions, ObservationModes
from stratego_env_amd.config import VARIANTS
from tests.helpers import GOLDEN, _load_npz, oracle_cvariant
from tests.test_gpu_parity import _table
from tests.test_oracle_golden_both import digest_both

pytestmark = pytest.mark.gpu
MASK, POBS, FOBS = 'valid_actions_mask', 'partial_observation', 'full_observation'


@pytest.mark.parametrize('name,n_envs,n_steps', [('barrage', 32, 500), ('standard', 6, 300), ('tiny', 48, 150), ('micro', 48, 80),
                                                 ('fives', 32, 150), ('octa_barrage', 16, 200), ('standard2', 2, 80)])
def test_full_obs_bit_exact_vs_oracle(name, n_envs, n_steps):
    check_both_obs_vs_oracle(name, n_envs, n_steps, 'extended')


def check_both_obs_vs_oracle(name, n_envs, n_steps, channel_mode):
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[name]
    seed, g0 = 0x77AA55 + len(name), 500
    env = VecStrategoEnv(name, n_envs, seed=seed, env_id_offset=g0, auto_reset=True, final_obs=True, full_obs=True,
                         obs_channel_mode=channel_mode)
    cv = oracle_cvariant(name, setups=_table(name))
    oenvs = []
    for e in range(n_envs):
        oe = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts,
                           observation_mode='both_observations', obs_channel_mode=channel_mode)
        oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, 0))
        oe.game_no = 0
        oenvs.append(oe)
    env.reset()
    fobs_h, obs_h, mask_h = env.fobs.cpu().numpy(), env.obs.cpu().numpy(), env.mask.cpu().numpy()
    cur = []
    for e, oe in enumerate(oenvs):
        o = oe._obs(1)
        assert o[FOBS].tobytes() == fobs_h[e].tobytes() and o[POBS].tobytes() == obs_h[e].tobytes()
        cur.append(o)
    env.sample_valid_actions()
    ended = 0
    for t in range(n_steps):
        acts = env.next_actions.cpu().numpy().copy()
        env.rollout_step()
        fobs_h, obs_h, mask_h = env.fobs.cpu().numpy(), env.obs.cpu().numpy(), env.mask.cpu().numpy()
        done_h, ff_h, fp_h = env.done.cpu().numpy(), env.final_fobs.cpu().numpy(), env.final_obs.cpu().numpy()
        for e, oe in enumerate(oenvs):
            o, rew, done, info = oe.step({oe.player: int(acts[e])})
            assert bool(done_h[e]) == done['__all__']
            if done['__all__']:
                ended += 1
                for slot, p in ((0, 1), (1, -1)):
                    assert o[p][FOBS].tobytes() == ff_h[e, slot].tobytes(), (name, t, e, 'final full obs', p)
                    assert o[p][POBS].tobytes() == fp_h[e, slot].tobytes()
                oe.game_no += 1
                o = oe.reset(initial_state_override=orc.reset_state(cv, seed, g0 + e, oe.game_no))
            p = oe.player
            assert o[p][FOBS].tobytes() == fobs_h[e].tobytes(), (name, t, e, 'full obs')
            assert o[p][POBS].tobytes() == obs_h[e].tobytes(), (name, t, e, 'partial obs')
            assert np.array_equal(o[p][MASK], mask_h[e])
    assert ended > 0 or name in ('standard', 'standard2')
    env.close()


@pytest.mark.parametrize('name', ['barrage', 'tiny', 'micro', 'fives'])
def test_facade_both_mode_replays_reference_goldens(name):
    """Default reference config (observation_mode = BOTH_OBSERVATIONS): keys and bytes of both observations."""
    from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv, DEFAULT_CONFIG
    from tests.test_gpu_facade import _state_from_maps
    assert DEFAULT_CONFIG['observation_mode'] == ObservationModes.BOTH_OBSERVATIONS
    g = _load_npz(os.path.join(GOLDEN, 'games_both_%s.npz' % name))
    env = StrategoMultiAgentEnv({'version': GameVersions(name)})        # no observation_mode: the default, BOTH
    off = g['offsets']
    for gi in range(min(5, len(off) - 1)):
        obs = env.reset(initial_state_override=_state_from_maps(name, g['p1_maps'][gi], g['p2_maps'][gi]))
        assert sorted(obs[1].keys()) == [FOBS, POBS, MASK]
        assert obs[1][FOBS].shape == (env.rows, env.columns, 79) and obs[1][FOBS].dtype == np.float32
        assert digest_both(obs) == int(g['init_digests'][gi])
        for k in range(off[gi], off[gi + 1]):
            obs, rew, done, info = env.step({env.player: int(g['actions'][k])})
            assert digest_both(obs) == int(g['digests'][k]), (name, gi, k)
            assert done['__all__'] == bool(g['dones'][k])
    env.close()
    env = StrategoMultiAgentEnv({'version': GameVersions(name), 'observation_mode': ObservationModes.FULLY_OBSERVABLE})
    obs = env.reset()
    assert sorted(obs[1].keys()) == [FOBS, MASK] and sorted(env.observation_space.spaces.keys()) == [FOBS, MASK]
    env.close()
