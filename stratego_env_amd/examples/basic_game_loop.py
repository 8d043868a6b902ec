"""Drop-in counterpart of the reference's examples/basic_game_loop.py (lines 6-63): the same loop body runs
unchanged against this package's StrategoMultiAgentEnv.

    python -m stratego_env_amd.examples.basic_game_loop [--games N] [--version standard]

The chooser mirrors `nnet_choose_action_example` (uniform logits, invalid actions masked to -inf, softmax over the
flattened (rows x cols x ways_to_move) mask, np.random.choice); `softmax` restates examples/util.py:4-47.
"""
import argparse

import numpy as np

from stratego_env_amd import GameVersions, ObservationComponents, ObservationModes
from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv


def softmax(x, temperature=1.0):
    x = np.asarray(x, dtype=np.float64) / temperature
    x = x - np.max(x)
    e = np.exp(x)
    return e / np.sum(e)


def nnet_choose_action_example(current_player, obs_from_env):
    board_observation = obs_from_env[current_player][ObservationComponents.PARTIAL_OBSERVATION.value]  # noqa: F841
    valid_actions_mask = obs_from_env[current_player][ObservationComponents.VALID_ACTIONS_MASK.value]
    logits = np.ones_like(valid_actions_mask, dtype=np.float64)
    neg_inf_mask = np.maximum(np.log(valid_actions_mask + 1e-8), np.finfo(np.float32).min)
    flat = np.reshape(logits + neg_inf_mask, -1)
    return np.random.choice(range(len(flat)), p=softmax(flat))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games', type=int, default=1)
    ap.add_argument('--version', default='standard')
    args = ap.parse_args()
    config = {
        'version': GameVersions(args.version),
        'random_player_assignment': True,
        'human_inits': args.version in ('standard', 'barrage', 'short_standard', 'medium_standard', 'short_barrage'),
        'observation_mode': ObservationModes.PARTIALLY_OBSERVABLE,
    }
    env = StrategoMultiAgentEnv(env_config=config)
    for _ in range(args.games):
        print("New Game Started")
        obs = env.reset()
        steps = 0
        while True:
            assert len(obs.keys()) == 1
            current_player = list(obs.keys())[0]
            assert current_player == 1 or current_player == -1
            action = nnet_choose_action_example(current_player, obs)
            obs, rew, done, info = env.step(action_dict={current_player: action})
            steps += 1
            if done["__all__"]:
                print(f"Game Finished after {steps} steps, player 1 rew: {rew[1]}, player -1 rew: {rew[-1]}")
                break
            else:
                assert all(r == 0.0 for r in rew.values())


if __name__ == '__main__':
    main()
