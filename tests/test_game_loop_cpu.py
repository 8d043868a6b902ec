"""The caller loop without a GPU: the example's policy (stratego_env_amd/examples/basic_game_loop.py) and the facade's reset-time
RNG consumption must reproduce the action sequence the REFERENCE's own loop chose (tests/golden/game_loop.json, recorded by
tools/oracle/gen_golden_game_loop.py), with the oracle standing in for the env.  The GPU test of the same golden
(tests/test_gpu_facade.py) then runs the product facade end to end."""
import hashlib
import json
import os
import random

import numpy as np

from oracle import oracle as orc
from stratego_env_amd import setups as S
from stratego_env_amd.config import VARIANTS
from tests.helpers import GOLDEN

MASK, POBS = 'valid_actions_mask', 'partial_observation'


def _digest(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(np.ascontiguousarray(obs[p][MASK]).astype(np.uint8).tobytes())
        h.update(np.ascontiguousarray(obs[p][POBS]).astype(np.float32).tobytes())
    return int.from_bytes(h.digest()[:8], 'little')


def test_example_policy_and_reset_draws_reproduce_the_reference_loop():
    from stratego_env_amd.examples import basic_game_loop as ex
    gold = json.load(open(os.path.join(GOLDEN, 'game_loop.json')))
    v = VARIANTS['standard']
    table = S.load_setup_table(v.human_inits)
    for g in gold['main_config'][:3]:
        np.random.seed(g['seed'])
        random.seed(g['seed'])
        swap = not (np.random.random() < 0.5)                       # random_player_assignment (maenv:537-543)
        m1, m2 = S.sample_initial_maps_like_reference(v, table)
        env = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts)
        obs = env.reset(m1, m2)
        relabel = (lambda d: {-k if k in (1, -1) else k: x for k, x in d.items()}) if swap else (lambda d: d)
        obs = relabel(obs)
        assert list(obs.keys()) == [g['first_key']] and _digest(obs) == g['init_digest']
        for k, want in enumerate(g['actions']):
            (mover,) = obs.keys()
            a = int(ex.nnet_choose_action_example(mover, {mover: {MASK: obs[mover][MASK].astype(np.int64), POBS: obs[mover][POBS]}}))
            assert a == want, (g['seed'], k)
            obs, rew, done, info = env.step({(-mover if swap else mover): a})
            obs = relabel(obs)
            assert _digest(obs) == g['digests'][k], (g['seed'], k)
        assert done['__all__'] and {str(p): float(r) for p, r in relabel(rew).items()} == g['rewards']
