"""Goldens for the remaining StrategoMultiAgentEnv options, recorded from the REFERENCE (BUILD CONTAINER ONLY):
repeat_games_from_other_side (maenv:530-534), penalize_ties (maenv:803-805), observation_includes_internal_state
(maenv:494-495), same_start_pos_everytime (maenv:352-354), reset(first_player_override=...) (maenv:555-558).
For each case: np.random.seed / random.seed, several reset + play-to-the-end episodes with the deterministic action rule
a = valid[(7919 * t) % len(valid)]; per reset and per step the returned keys, a digest of every observation component,
rewards, dones and infos.  Output: tests/golden/facade_options.json
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


def obs_digest(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        for comp in sorted(obs[p].keys()):
            a = np.asarray(obs[p][comp])
            a = a.astype(np.uint8) if comp == 'valid_actions_mask' else a.astype(np.int64) if comp == 'internal_state' else a
            h.update(comp.encode())
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


CASES = [
    dict(name='repeat_other_side', cfg={'version': 'tiny', 'repeat_games_from_other_side': True}, episodes=4, seed=3),
    dict(name='penalize_ties', cfg={'version': 'micro', 'penalize_ties': True}, episodes=12, seed=5),
    dict(name='internal_state', cfg={'version': 'tiny', 'observation_includes_internal_state': True,
                                      'observation_mode': 'both_observations'}, episodes=2, seed=8),
    dict(name='same_start', cfg={'version': 'fives', 'same_start_pos_everytime': True, 'random_player_assignment': True},
         episodes=3, seed=13),
    dict(name='first_player_override', cfg={'version': 'tiny'}, episodes=3, seed=21, first_player_override=-1),
    dict(name='human_same_start_relabel', cfg={'version': 'short_barrage', 'human_inits': True, 'same_start_pos_everytime': True,
                                               'random_player_assignment': True}, episodes=3, seed=55),
    dict(name='override_consumes_rng', cfg={'version': 'tiny'}, episodes=3, seed=89, override_second_reset=True),
    dict(name='actions_1d_oscillation', cfg={'version': 'tiny'}, episodes=2, seed=144, one_dim=True),
    dict(name='barrage_human_repeat', cfg={'version': 'barrage', 'human_inits': True, 'repeat_games_from_other_side': True},
         episodes=2, seed=34, max_steps=60),
]


def main():
    ref = import_reference()
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    out = []
    for case in CASES:
        cfg = dict(case['cfg'])
        cfg['version'] = GV(cfg['version'])
        cfg['observation_mode'] = OM(cfg.get('observation_mode', 'partially_observable'))
        np.random.seed(case['seed'])
        random.seed(case['seed'])
        env = ref.maenv.StrategoMultiAgentEnv(cfg)
        eps = []
        first_state = None
        for e_i in range(case['episodes']):
            override = first_state if (case.get('override_second_reset') and e_i == 1) else None
            obs = env.reset(first_player_override=case.get('first_player_override'), initial_state_override=override)
            if first_state is None:
                first_state = np.array(env.state, copy=True)
            ep = dict(keys=sorted(int(k) for k in obs), comps=sorted(list(obs.values())[0].keys()), player=int(env.player),
                      init=obs_digest(obs), steps=[])
            t = 0
            while True:
                k = list(obs.keys())[0]
                valid = np.flatnonzero(obs[k]['valid_actions_mask'].reshape(-1))
                a = int(valid[(7919 * t) % len(valid)])
                if case.get('one_dim'):      # the same move as a 1-D index in the mover's perspective, two-square rule off
                    a1 = int(env.base_env.get_action_1d_index_from_spatial_index(np.unravel_index(a, env.base_env.spatial_action_size)))
                    obs, rew, done, info = env.step({k: a1}, is_spatial_index=False, allow_piece_oscillation=True)
                else:
                    obs, rew, done, info = env.step({k: a})
                ep['steps'].append(dict(a=a, keys=sorted(int(x) for x in obs), d=obs_digest(obs), done=bool(done['__all__']),
                                        rew={str(kk): float(vv) for kk, vv in rew.items()},
                                        info={str(kk): vv for kk, vv in info.items()}))
                t += 1
                if done['__all__'] or t >= case.get('max_steps', 10 ** 9):
                    break
            eps.append(ep)
        out.append(dict(name=case['name'], cfg=case['cfg'], seed=case['seed'], first_player_override=case.get('first_player_override'),
                        override_second_reset=bool(case.get('override_second_reset')), one_dim=bool(case.get('one_dim')),
                        max_steps=case.get('max_steps'), episodes=eps))
        print(case['name'], [len(e['steps']) for e in eps])
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'facade_options.json'), 'w'))


if __name__ == '__main__':
    main()
