"""Oracle in observation_mode=BOTH_OBSERVATIONS vs vectors recorded from the reference (tools/oracle/gen_golden_both.py)."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as orc
from stratego_env_amd.config import VARIANTS
from tests.helpers import _load_npz, GOLDEN, load_variants_json
import os


def digest_both(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(np.ascontiguousarray(obs[p]['valid_actions_mask']).astype(np.uint8).tobytes())
        h.update(np.ascontiguousarray(obs[p]['partial_observation'], dtype=np.float32).tobytes())
        h.update(np.ascontiguousarray(obs[p]['full_observation'], dtype=np.float32).tobytes())
    return int.from_bytes(h.digest()[:8], 'little')


def test_full_obs_norm_constants_match_reference():
    ref = load_variants_json()['variants']
    for name, r in ref.items():
        mids, ranges = orc.f_obs_norm_constants(VARIANTS[name].piece_counts)
        assert np.array_equal(mids, np.asarray(r['f_obs_mids'], dtype=np.float32)), name
        assert np.array_equal(ranges, np.asarray(r['f_obs_ranges'], dtype=np.float32)), name


@pytest.mark.parametrize('name', ['barrage', 'standard', 'tiny', 'micro', 'fives'])
def test_replay_both_mode_goldens(name):
    g = _load_npz(os.path.join(GOLDEN, 'games_both_%s.npz' % name))
    v = VARIANTS[name]
    env = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts, observation_mode='both_observations')
    off = g['offsets']
    for gi in range(len(off) - 1):
        obs = env.reset(g['p1_maps'][gi].astype(np.int64), g['p2_maps'][gi].astype(np.int64))
        assert sorted(obs[1].keys()) == ['full_observation', 'partial_observation', 'valid_actions_mask']
        assert digest_both(obs) == int(g['init_digests'][gi])
        for k in range(off[gi], off[gi + 1]):
            obs, rew, done, info = env.step({env.player: int(g['actions'][k])})
            assert digest_both(obs) == int(g['digests'][k]), (name, gi, k)
            assert bool(done['__all__']) == bool(g['dones'][k])
