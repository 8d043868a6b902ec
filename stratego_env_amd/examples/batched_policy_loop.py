"""The reference's game loop (examples/basic_game_loop.py:34-63) for N concurrent games with a policy on the GPU.

    python -m stratego_env_amd.examples.batched_policy_loop [--games 65536] [--steps 200] [--version barrage]

`nnet_choose_action_example` (basic_game_loop.py:6-31) for a batch: logits over the flattened (rows x cols x ways_to_move)
action space (here a fixed random linear read-out of the observation, standing in for a network), invalid actions masked to
-inf, softmax, one sample per game -- all on the device; the chosen flat indices go straight back into `step`.  Finished games
restart inside the step (auto_reset), so the loop never leaves the GPU.  Prints steps/s and per-player win counts.
"""
import argparse
import time

import torch

from stratego_env_amd.vec_env import VecStrategoEnv


def policy_logits(obs, readout, out=None):
    """The stand-in for a network: obs float32 [N,R,C,67] -> logits float32 [N, R*C*K] (mean over the board, then a fixed linear read-out)."""
    feat = obs.mean(dim=(1, 2))                                   # [N, 67]
    return torch.matmul(feat, readout, out=out)                   # [N, R*C*K]


def choose_actions(obs, mask, readout, generator):
    """obs float32 [N,R,C,67], mask uint8 [N,R,C,K] -> int32 [N] flat action indices of the current movers, composed from torch ops (the
    round-4 consumer: masked_fill, softmax and multinomial make four more passes over [N, R*C*K])."""
    n = obs.shape[0]
    logits = policy_logits(obs, readout).view(n, -1)
    logits = logits.masked_fill(mask.view(n, -1) == 0, float('-inf'))
    probs = torch.softmax(logits, dim=1)
    return torch.multinomial(probs, 1, generator=generator).view(n).to(torch.int32)


def choose_actions_fused(env, obs, readout, temperature=1.0, logits_out=None):
    """The same policy with the library's chooser (sgx_choose_actions): the logits are read once, together with the mask the step kernel
    wrote, and one action per game is drawn on the device with the env's counter RNG -- reproducible from (seed, env id, game, turn)."""
    logits = policy_logits(obs, readout, out=logits_out)
    return env.choose_actions(logits, temperature)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--version', default='barrage')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--torch-chooser', action='store_true', help='mask / softmax / sample composed from torch ops instead of sgx_choose_actions')
    args = ap.parse_args()
    env = VecStrategoEnv(args.version, args.games, seed=args.seed, auto_reset=True)
    obs, mask, player = env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(args.seed)
    readout = torch.randn(obs.shape[-1], mask[0].numel(), device=env.device, generator=g) * 0.5
    wins = torch.zeros(2, dtype=torch.int64, device=env.device)
    finished = torch.zeros((), dtype=torch.int64, device=env.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logits_buf = torch.empty((args.games, mask[0].numel()), dtype=torch.float32, device=env.device)
    for _ in range(args.steps):
        actions = choose_actions(obs, mask, readout, g) if args.torch_chooser else choose_actions_fused(env, obs, readout, logits_out=logits_buf)
        obs, mask, reward, done, player = env.step(actions)
        wins += (reward > 0).sum(dim=0)
        finished += done.sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(env.invalid_action.sum()) == 0
    print("%d %s games x %d steps with a device-side policy: %.1f M env steps/s; %d games finished, wins +1: %d, -1: %d" %
          (args.games, args.version, args.steps, args.games * args.steps / dt / 1e6, int(finished), int(wins[0]), int(wins[1])))
    env.close()


if __name__ == '__main__':
    main()
