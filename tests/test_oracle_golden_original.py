"""Oracle with obs_channel_mode='original' (32/33-layer observations) vs vectors recorded from the reference
(tools/oracle/gen_golden_original.py)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd.config import VARIANTS
from tests.helpers import _load_npz, GOLDEN
from tests.test_oracle_golden_both import digest_both

ORIG_VARIANTS = ['barrage', 'standard', 'tiny', 'micro', 'fives', 'octa_barrage']


def test_original_norm_constants_match_reference():
    with open(os.path.join(GOLDEN, 'orig_norm.json')) as f:
        ref = json.load(f)
    for name, r in ref.items():
        pm, pr = orc.p_obs_norm_constants(VARIANTS[name].piece_counts, original=True)
        fm, fr = orc.f_obs_norm_constants(VARIANTS[name].piece_counts, original=True)
        assert pm.shape == (32,) and fm.shape == (33,)
        assert np.array_equal(pm, np.asarray(r['p_obs_mids'], dtype=np.float32)), name
        assert np.array_equal(pr, np.asarray(r['p_obs_ranges'], dtype=np.float32)), name
        assert np.array_equal(fm, np.asarray(r['f_obs_mids'], dtype=np.float32)), name
        assert np.array_equal(fr, np.asarray(r['f_obs_ranges'], dtype=np.float32)), name


@pytest.mark.parametrize('name', ORIG_VARIANTS)
def test_replay_original_mode_goldens(name):
    g = _load_npz(os.path.join(GOLDEN, 'games_orig_%s.npz' % name))
    v = VARIANTS[name]
    env = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts,
                        observation_mode='both_observations', obs_channel_mode='original')
    off = g['offsets']
    for gi in range(len(off) - 1):
        obs = env.reset(g['p1_maps'][gi].astype(np.int64), g['p2_maps'][gi].astype(np.int64))
        assert obs[1]['partial_observation'].shape == (v.rows, v.columns, 32)
        assert obs[1]['full_observation'].shape == (v.rows, v.columns, 33)
        assert digest_both(obs) == int(g['init_digests'][gi])
        for k in range(off[gi], off[gi + 1]):
            obs, rew, done, info = env.step({env.player: int(g['actions'][k])})
            assert digest_both(obs) == int(g['digests'][k]), (name, gi, k)
            assert bool(done['__all__']) == bool(g['dones'][k])
