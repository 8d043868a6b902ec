"""Per-game cost of the step kernel by batch size and variant (is a slow big batch a placement or an instruction problem?).

    python tools/size_sweep.py [--steps 128]

For each (variant, games): plain torch allocation, `steps` rollout steps after a warm-up that plays games into their middle
phase; prints us per launch, ns per game, algorithmic TB/s.  Also the same with observation / mask outputs switched off.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import b_alg  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn(steps)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--warm', type=int, default=200)
    ap.add_argument('--specs', default='barrage:65536,barrage:262144,standard:65536,standard:262144,barrage:16384,standard:16384')
    args = ap.parse_args()
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    for spec in args.specs.split(','):
        name, n = spec.split(':')
        n = int(n)
        env = VecStrategoEnv(name, n, seed=0x5712A7E60, auto_reset=True)
        env.reset()
        env.rollout_steps(args.warm)
        torch.cuda.synchronize()
        full = min(timed(env.rollout_steps, args.steps) for _ in range(3))

        def no_out(k):
            for _ in range(k):
                env.step(env.next_actions, want_next_actions=True, emit_obs=False, emit_mask=False)
        logic = min(timed(no_out, args.steps) for _ in range(2))

        def no_obs(k):
            for _ in range(k):
                env.step(env.next_actions, want_next_actions=True, emit_obs=False, emit_mask=True)
        noobs = min(timed(no_obs, args.steps) for _ in range(2))
        v = env.variant
        print("%-10s %7d games: %8.1f us/launch  %6.2f ns/game  %5.2f TB/s alg (frac %.3f) | no obs %7.1f us | no obs+mask %7.1f us" %
              (name, n, full, full * 1e3 / n, b_alg(v.rows, v.columns) * n / full / 1e6, b_alg(v.rows, v.columns) * n / full / 1e6 / 8.0,
               noobs, logic), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
