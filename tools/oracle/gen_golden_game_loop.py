"""Golden vectors of the reference's own caller loop (container only): examples/basic_game_loop.py's __main__ configuration --
STANDARD, human_inits, random_player_assignment, PARTIALLY_OBSERVABLE -- driven by the reference's own
nnet_choose_action_example after np.random.seed(s); random.seed(s).

Stored per game: seed, the key of the first observation, every action the loop chose, the digest of every observation dict
step() returned (keys as returned, i.e. after the random relabelling), the terminal rewards / infos.  A second block records
the same loop on the reference's DEFAULT observation mode (BOTH_OBSERVATIONS) for two short-variant games.
Output: tests/golden/game_loop.json
"""
import hashlib
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402


def digest(obs, keys=('valid_actions_mask', 'partial_observation')):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        for k in keys:
            if k in obs[p]:
                a = np.ascontiguousarray(obs[p][k])
                h.update((a.astype(np.uint8) if k == 'valid_actions_mask' else a.astype(np.float32)).tobytes())
    return int.from_bytes(h.digest()[:8], 'little')


def play(ref, loop, cfg, seed, keys):
    np.random.seed(seed)
    random.seed(seed)
    env = ref.maenv.StrategoMultiAgentEnv(env_config=cfg)
    obs = env.reset()
    rec = dict(seed=seed, first_key=int(list(obs.keys())[0]), init_digest=digest(obs, keys), actions=[], digests=[])
    while True:
        assert len(obs.keys()) == 1
        p = list(obs.keys())[0]
        a = int(loop.nnet_choose_action_example(current_player=p, obs_from_env=obs))
        obs, rew, done, info = env.step(action_dict={p: a})
        rec['actions'].append(a)
        rec['digests'].append(digest(obs, keys))
        if done['__all__']:
            rec['rewards'] = {str(k): float(v) for k, v in rew.items()}
            rec['infos'] = {str(k): {kk: (bool(vv) if isinstance(vv, (bool, np.bool_)) else vv) for kk, vv in v.items()}
                            for k, v in info.items()}
            break
        assert all(r == 0.0 for r in rew.values())
    return rec


def main():
    ref = import_reference()
    import importlib
    loop = importlib.import_module('stratego_env.examples.basic_game_loop')
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    main_cfg = {'version': GV.STANDARD, 'random_player_assignment': True, 'human_inits': True,
                'observation_mode': OM.PARTIALLY_OBSERVABLE}
    out = {'main_config': [play(ref, loop, dict(main_cfg), s, ('valid_actions_mask', 'partial_observation')) for s in (0, 1, 2, 3)]}
    both = ('valid_actions_mask', 'partial_observation', 'full_observation')
    out['default_mode'] = []
    for name, seed in (('short_standard', 11), ('barrage', 12)):
        cfg = {'version': GV(name), 'random_player_assignment': True, 'human_inits': True}     # observation_mode: the default (BOTH)
        r = play(ref, loop, cfg, seed, both)
        r['version'] = name
        out['default_mode'].append(r)
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'game_loop.json'), 'w'))
    for k, v in out.items():
        print(k, [(g['seed'], len(g['actions']), g['rewards']) for g in v])


if __name__ == '__main__':
    main()
