"""In-process A/B of the store policy of the observation writes on ONE env object (same buffers, same games):
sgx_set_nt_stores(0 | 1) alternating, at a given depth into the games (Standard's uncoded captured counts only appear mid-game).

    python tools/nt_ab.py [standard:262144:300,standard:65536:300,barrage:262144:300]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn(steps)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    specs = sys.argv[1] if len(sys.argv) > 1 else 'standard:262144:300,standard:65536:300,barrage:262144:300'
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    for spec in specs.split(','):
        name, n, warm = spec.split(':')
        n, warm = int(n), int(warm)
        env = VecStrategoEnv(name, n, seed=0x5712A7E60, auto_reset=True)
        env.reset()
        env.rollout_steps(warm)
        torch.cuda.synchronize()
        best = {0: 1e9, 1: 1e9}
        for rnd in range(4):
            for mode in (0, 1):
                env.set_nt_stores(bool(mode))
                best[mode] = min(best[mode], timed(env.rollout_steps, 48))
        print("%-10s %7d games, %4d steps in: plain stores %8.1f us   non-temporal %8.1f us  (%+.1f %%)" %
              (name, n, warm, best[0], best[1], 100 * (best[1] / best[0] - 1)), flush=True)
        env.close()
        del env


if __name__ == '__main__':
    main()
