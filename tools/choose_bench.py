"""sgx_choose_actions alone: us per call and TB/s on 65,536 Barrage games (random logits, the env's current mask).
    python tools/choose_bench.py [barrage] [65536]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    env = VecStrategoEnv(version, n, seed=1, auto_reset=True)
    env.reset()
    env.rollout_steps(30)
    na = env.R * env.Cc * env.K
    logits = torch.randn((n, na), device=env.device)
    out = torch.empty((n,), dtype=torch.int32, device=env.device)
    for temp in (1.0, 0.0):
        for _ in range(5):
            env.choose_actions(logits, temp, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            env.choose_actions(logits, temp, out=out)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 50 * 1e3
        byts = n * (4 * na + na + 36)
        print("%s %d games, temperature %.1f: %.1f us per call = %.2f TB/s (%.3f of 8 TB/s) on %d B per game" % (version, n, temp, us, byts / us / 1e6, byts / us / 1e6 / 8, 4 * na + na + 36))
    env.close()


if __name__ == '__main__':
    main()
