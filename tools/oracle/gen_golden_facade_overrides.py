"""Goldens for env_config dicts that OVERRIDE fields of the version's config (the reference merges env_config over
VERSION_CONFIGS[version], maenv:320-323), recorded from the REFERENCE (BUILD CONTAINER ONLY).  Which overrides take effect is the
reference's business (maenv:325-349): piece_amounts -> normalisation only; max_turns / obstacle_locations -> only with human_inits;
initial_state_usable_rows -> nothing.  Same recording as gen_golden_facade_options.py.  Output: tests/golden/facade_overrides.json
(piece_amounts overrides are stored with the SP member NAMES as keys).
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402
from tools.oracle.gen_golden_facade_options import obs_digest  # noqa: E402

CASES = [
    dict(name='barrage_human_max_turns_40', cfg={'version': 'barrage', 'human_inits': True, 'max_turns': 40}, episodes=3, seed=2),
    dict(name='tiny_random_max_turns_ignored', cfg={'version': 'tiny', 'max_turns': 6}, episodes=3, seed=4),
    dict(name='standard_human_piece_amounts', cfg={'version': 'standard', 'human_inits': True, 'observation_mode': 'both_observations',
                                                   'piece_amounts': {'SPY': 1, 'SCOUT': 4, 'MINER': 2, 'SERGEANT': 4, 'LIEUTENANT': 3,
                                                                     'CAPTAIN': 4, 'MAJOR': 6, 'COLONEL': 2, 'GENERAL': 1, 'MARSHALL': 1,
                                                                     'FLAG': 1, 'BOMB': 5}},
         episodes=1, seed=6, max_steps=260),
    dict(name='short_barrage_human_obstacles', cfg={'version': 'short_barrage', 'human_inits': True,
                                                    'obstacle_locations': [(4, 4), (5, 5), (4, 5), (5, 0)]}, episodes=2, seed=8),
    dict(name='micro_piece_amounts_norm_only', cfg={'version': 'micro', 'piece_amounts': {'LIEUTENANT': 2, 'CAPTAIN': 3, 'FLAG': 1},
                                                    'observation_mode': 'both_observations'}, episodes=6, seed=10),
    dict(name='barrage_random_obstacles_ignored', cfg={'version': 'short_barrage', 'obstacle_locations': [(4, 4)], 'max_turns': 30},
         episodes=1, seed=12),
    dict(name='fives_usable_rows_ignored', cfg={'version': 'fives', 'initial_state_usable_rows': 2, 'obs_channel_mode': 'original'},
         episodes=2, seed=14),
    dict(name='short_standard_human_all', cfg={'version': 'short_standard', 'human_inits': True, 'max_turns': 120,
                                               'obstacle_locations': [(4, 1), (5, 8)], 'random_player_assignment': True,
                                               'piece_amounts': {'SPY': 1, 'SCOUT': 8, 'MINER': 5, 'SERGEANT': 4, 'LIEUTENANT': 4,
                                                                 'CAPTAIN': 4, 'MAJOR': 3, 'COLONEL': 2, 'GENERAL': 2, 'MARSHALL': 1,
                                                                 'FLAG': 1, 'BOMB': 6}},
         episodes=2, seed=16),
]


def main():
    ref = import_reference()
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    SP = ref.impl.SP
    out = []
    for case in CASES:
        cfg = dict(case['cfg'])
        cfg['version'] = GV(cfg['version'])
        cfg['observation_mode'] = OM(cfg.get('observation_mode', 'partially_observable'))
        if 'piece_amounts' in cfg:
            cfg['piece_amounts'] = {SP[k]: n for k, n in cfg['piece_amounts'].items()}
        np.random.seed(case['seed'])
        random.seed(case['seed'])
        env = ref.maenv.StrategoMultiAgentEnv(cfg)
        eps = []
        for e_i in range(case['episodes']):
            obs = env.reset()
            ep = dict(keys=sorted(int(k) for k in obs), comps=sorted(list(obs.values())[0].keys()), player=int(env.player),
                      init=obs_digest(obs), steps=[], max_turns_in_state=int(env.state[5, 1, 0]),
                      obstacles=[[int(r), int(c)] for r, c in zip(*np.nonzero(env.state[2]))])
            t = 0
            while True:
                k = list(obs.keys())[0]
                valid = np.flatnonzero(obs[k]['valid_actions_mask'].reshape(-1))
                a = int(valid[(7919 * t) % len(valid)])
                obs, rew, done, info = env.step({k: a})
                ep['steps'].append(dict(a=a, keys=sorted(int(x) for x in obs), d=obs_digest(obs), done=bool(done['__all__']),
                                        rew={str(kk): float(vv) for kk, vv in rew.items()},
                                        info={str(kk): vv for kk, vv in info.items()}))
                t += 1
                if done['__all__'] or t >= case.get('max_steps', 10 ** 9):
                    break
            eps.append(ep)
        out.append(dict(name=case['name'], cfg=case['cfg'], seed=case['seed'], max_steps=case.get('max_steps'), episodes=eps))
        print(case['name'], [len(e['steps']) for e in eps], [e['max_turns_in_state'] for e in eps])
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'facade_overrides.json'), 'w'))


if __name__ == '__main__':
    main()
