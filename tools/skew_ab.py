"""In-process A/B of the XCD shares (sgx_set_xcd_skew) on one env and one set of output buffers: launch time per board size / batch
size for several per-mille values, interleaved rounds.    python tools/skew_ab.py [spec,spec,...]   spec = variant:games"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

SKEWS = (0, 60, 100, 140, 200)


def main():
    specs = (sys.argv[1] if len(sys.argv) > 1 else
             'barrage:65536,barrage:262144,standard:262144,octa_barrage:65536,medium:65536,fives:65536,micro:262144,tiny:262144,standard2:32768,micro:65536')
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    for spec in specs.split(','):
        name, n = spec.split(':')
        n = int(n)
        env = VecStrategoEnv(name, n, seed=0x5712A7E60, auto_reset=True)
        env.reset()
        env.rollout_steps(100)
        times = {s: [] for s in SKEWS}
        steps = 48
        for r in range(5):
            for s in SKEWS:
                env.set_xcd_skew(s)
                env.rollout_steps(4)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                env.rollout_steps(steps)
                b.record()
                torch.cuda.synchronize()
                times[s].append(a.elapsed_time(b) / steps * 1e3)
        base = statistics.median(times[0])
        print("%-13s %7d games: " % (name, n) + "  ".join("%d: %.1f us (%+.1f%%)" % (s, statistics.median(times[s]), 100 * (statistics.median(times[s]) / base - 1)) for s in SKEWS), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
