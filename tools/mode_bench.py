"""Launch time of the step kernel in the non-default output modes (full observation, 'original' channels) and variants.

    python tools/mode_bench.py [version envs steps]...      e.g.  barrage 65536 128  standard2 32768 64
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def run(version, n, steps, full, mode):
    env = VecStrategoEnv(version, n, seed=7, auto_reset=True, full_obs=full, obs_channel_mode=mode)
    env.reset()
    env.tune_placement()
    env.sample_valid_actions()
    env.rollout_steps(16)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.rollout_steps(steps)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / steps * 1e3
    v = env.variant
    out_bytes = 4 * v.rows * v.columns * (env.p_channels + (env.f_channels if full else 0)) + v.rows * v.columns * env.K
    print("%-14s %8d games  %-8s %-5s %9.1f us/launch %8.1f M steps/s  outputs %.2f TB/s" %
          (version, n, mode, 'both' if full else 'part', us, n / us, out_bytes * n / us / 1e6), flush=True)
    env.close()


if __name__ == '__main__':
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    args = sys.argv[1:] or ['barrage', '65536', '128']
    for i in range(0, len(args), 3):
        ver, n, steps = args[i], int(args[i + 1]), int(args[i + 2])
        for full in (False, True):
            for mode in ('extended', 'original'):
                run(ver, n, steps, full, mode)
