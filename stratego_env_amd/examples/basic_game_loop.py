"""Single-game loop through the drop-in dict API: what the reference's examples/basic_game_loop.py (lines 6-63) does, written
against this package's StrategoMultiAgentEnv.

    python -m stratego_env_amd.examples.basic_game_loop [--games N] [--version standard]

`nnet_choose_action_example` plays the role of the reference's function of that name (basic_game_loop.py:6-31): a policy
that puts equal logits on every action, sends invalid ones to -inf through the valid-actions mask, and samples one flat
(rows x cols x ways_to_move) index from the softmax (examples/util.py:4-47).
"""
import argparse

import numpy as np

from stratego_env_amd import GameVersions, ObservationComponents, ObservationModes
from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv

HUMAN_INIT_VERSIONS = ('standard', 'barrage', 'short_standard', 'medium_standard', 'short_barrage')
MASK_KEY = ObservationComponents.VALID_ACTIONS_MASK.value
OBS_KEY = ObservationComponents.PARTIAL_OBSERVATION.value


def softmax(x, temperature=1.0):
    z = np.asarray(x, dtype=np.float64) / temperature
    e = np.exp(z - z.max())
    return e / e.sum()


def nnet_choose_action_example(current_player, obs_from_env):
    mask = obs_from_env[current_player][MASK_KEY]
    _board = obs_from_env[current_player][OBS_KEY]       # a real policy network would read this
    masked_logits = np.ones(mask.shape, dtype=np.float64) + np.maximum(np.log(mask + 1e-8), np.finfo(np.float32).min)
    p = softmax(masked_logits.reshape(-1))
    return np.random.choice(p.shape[0], p=p)


def play_one_game(env, trace=None):
    """-> (steps, rewards dict of the terminal step).  `trace`: optional list that receives the reset observation and then
    one (action, obs, rewards, dones, infos) tuple per step (the parity tests compare it with the reference's own loop)."""
    obs, steps = env.reset(), 0
    if trace is not None:
        trace.append(obs)
    while True:
        (mover,) = obs.keys()                             # exactly one agent is asked to act
        assert mover in (1, -1)
        action = nnet_choose_action_example(mover, obs)
        obs, rewards, dones, infos = env.step(action_dict={mover: action})
        steps += 1
        if trace is not None:
            trace.append((int(action), obs, rewards, dones, infos))
        if dones["__all__"]:
            return steps, rewards
        assert all(r == 0.0 for r in rewards.values())


def make_env(version='standard', observation_mode=ObservationModes.PARTIALLY_OBSERVABLE):
    """The env of the reference's __main__ block (basic_game_loop.py:34-42): STANDARD, human setups, random player assignment."""
    return StrategoMultiAgentEnv(env_config={
        'version': GameVersions(version),
        'random_player_assignment': True,
        'human_inits': version in HUMAN_INIT_VERSIONS,
        'observation_mode': observation_mode,
    })


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games', type=int, default=1)
    ap.add_argument('--version', default='standard')
    args = ap.parse_args()
    env = make_env(args.version)
    for g in range(args.games):
        steps, rewards = play_one_game(env)
        print("game %d: %d steps, reward of player 1: %s, of player -1: %s" % (g, steps, rewards[1], rewards[-1]))
    env.close()


if __name__ == '__main__':
    main()
