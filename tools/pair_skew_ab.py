"""Do the speed classes of output buffers come from an affinity between device memory and XCD PAIRS that unequal shares can
compensate?  Held candidate observation buffers; per buffer the launch time with the pairs (0,1),(6,7) given (1 - d) and the pairs
(2,3),(4,5) given (1 + d) of the odd / even-skewed share, d = -0.2 .. +0.2.   python tools/pair_skew_ab.py [variant] [games] [buffers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

DS = (-0.2, -0.12, -0.06, 0.0, 0.06, 0.12, 0.2)


def timed(env, steps=16):
    env.rollout_steps(2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.rollout_steps(steps)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    env = VecStrategoEnv(name, n, seed=0x5712A7E60, auto_reset=True)
    env.reset()
    env.rollout_steps(100)
    bufs = [torch.empty_like(env.obs) for _ in range(nb)]
    print("d =        " + "  ".join("%+.2f" % d for d in DS))
    for i, buf in enumerate(bufs):
        env.obs = buf
        row = []
        for d in DS:
            w = []
            for xcd in range(8):
                pair_outer = xcd in (0, 1, 6, 7)
                base = 1100 if xcd % 2 == 0 else 900
                w.append(int(base * ((1 - d) if pair_outer else (1 + d))))
            env.set_xcd_shares(w)
            row.append(timed(env))
        print("buffer %2d: " % i + "  ".join("%5.1f" % t for t in row) + "   best d %+.2f (%.1f%% under d = 0)" %
              (DS[row.index(min(row))], 100 * (1 - min(row) / row[3])), flush=True)
    env.close()


if __name__ == '__main__':
    main()
