"""Golden vectors for the facade's reset-time RNG consumption, generated from the REFERENCE (container only).

For several (version, human_inits, seed): np.random.seed(seed); random.seed(seed); two consecutive env.reset() calls
-> the own-side piece maps of both games (recovered from env.state) and, with random_player_assignment, the key
under which the first observation is returned.   Output: tests/golden/facade_reset.json
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402


def main():
    ref = import_reference()
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    out = []
    for name, human in (('barrage', True), ('standard', True), ('barrage', False), ('tiny', False), ('micro', False),
                        ('octa_barrage', False)):
        for seed in (0, 1, 12345):
            np.random.seed(seed)
            random.seed(seed)
            env = ref.maenv.StrategoMultiAgentEnv({'version': GV(name), 'human_inits': human, 'random_player_assignment': True,
                                                   'observation_mode': OM.PARTIALLY_OBSERVABLE})
            games = []
            for _ in range(2):
                obs = env.reset()
                st = env.state
                games.append(dict(first_key=int(list(obs.keys())[0]), p1_map=st[0].tolist(), p2_map=st[1][::-1, ::-1].tolist()))
            out.append(dict(version=name, human_inits=human, seed=seed, games=games))
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'facade_reset.json'), 'w'))
    print(len(out), 'cases')


if __name__ == '__main__':
    main()
