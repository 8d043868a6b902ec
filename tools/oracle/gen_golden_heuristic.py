"""Golden for _get_heuristic_rewards_from_move (impl:852-891), recorded from the REFERENCE (BUILD CONTAINER ONLY) on the constructed
positions of tests.helpers.directed_positions with a fixed random reward matrix.  Output: tests/golden/heuristic_rewards.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402
from tests.helpers import directed_positions  # noqa: E402


def main():
    ref = import_reference()
    states, players, actions = directed_positions()
    rm = np.random.RandomState(2024).rand(14, 14).astype(np.float32)
    pe = ref.penv.StrategoProceduralEnv(4, 4)
    out = [float(ref.impl._get_heuristic_rewards_from_move(states[i], np.int64(players[i]), np.int64(actions[i]), pe.action_size,
                                                           pe._mpapsp, False, rm)) for i in range(len(states))]
    json.dump(dict(matrix_seed=2024, rewards=out), open(os.path.join(ROOT, 'tests', 'golden', 'heuristic_rewards.json'), 'w'))
    print(len(out), 'rewards;', sum(1 for x in out if x == 0), 'zero (no-op)')


if __name__ == '__main__':
    main()
