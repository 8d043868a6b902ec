"""Single-game latency of the drop-in facade: StrategoMultiAgentEnv.step() through the
basic_game_loop pattern (random valid action per step), steps per second on one host core + one GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratego_env_amd import GameVersions, ObservationModes  # noqa: E402
from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv  # noqa: E402


def run(mode, n_steps):
    env = StrategoMultiAgentEnv({'version': GameVersions.BARRAGE, 'human_inits': True, 'observation_mode': mode})
    np.random.seed(1)
    obs = env.reset()
    t0 = time.perf_counter()
    steps = games = 0
    while steps < n_steps:
        p = list(obs.keys())[0]
        valid = np.flatnonzero(obs[p]['valid_actions_mask'].reshape(-1))
        obs, rew, done, info = env.step({p: int(valid[np.random.randint(len(valid))])})
        steps += 1
        if done['__all__']:
            games += 1
            obs = env.reset()
    dt = time.perf_counter() - t0
    env.close()
    return steps / dt, games


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    for mode in (ObservationModes.PARTIALLY_OBSERVABLE, ObservationModes.BOTH_OBSERVATIONS):
        run(mode, 200)
        sps, games = run(mode, n)
        print("facade N=1 %-22s %8.0f steps/s (%d steps, %d games finished)" % (mode.value, sps, n, games))
