"""In-process interleaved A/B of alternative builds of the HIP library (cdna guide rule 24).

    python tools/ab_bench.py [--steps 64] [--rounds 7] lib1.so lib2.so ...

All variants play the same workload from the same state (each has its own handle / internal state tensor, all share
ONE set of output tensors), rounds are interleaved, and the median / min launch time per variant is printed.
"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--version', default='barrage')
    ap.add_argument('--tune', action='store_true', help='the shared output tensors come from the placement trial (tune_placement) instead of a plain allocation')
    ap.add_argument('--warm', type=int, default=32, help='steps played before the timed rounds (how deep into the games)')
    args = ap.parse_args()
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    envs = []
    for i, path in enumerate(args.libs):
        e = VecStrategoEnv(args.version, args.envs, seed=0x5712A7E60, auto_reset=True, lib_path=os.path.abspath(path))
        if envs:   # share the big output tensors
            e.obs, e.mask = envs[0].obs, envs[0].mask
        e.reset()
        if args.tune and not envs:
            rep = e.tune_placement()
            print("placement trial: first %.1f us, kept %.1f us, %d candidates" % (rep['obs'][0], min(rep['obs']), len(rep['obs'])))
        e.sample_valid_actions()
        e.rollout_steps(args.warm)
        envs.append(e)
    times = [[] for _ in envs]
    for r in range(args.rounds):
        for i, e in enumerate(envs):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.steps):
                e.rollout_step()
            b.record()
            torch.cuda.synchronize()
            times[i].append(a.elapsed_time(b) / args.steps * 1e3)
    for path, t in zip(args.libs, times):
        print("%-28s median %.1f us  min %.1f us  max %.1f us  -> %.1f M steps/s" %
              (os.path.basename(path), statistics.median(t), min(t), max(t), args.envs / statistics.median(t)))


if __name__ == '__main__':
    main()
