"""In-process A/B of wave priorities by SIMD slot (sgx_layout.h: stagger_priority, experiment) on the wave-per-game kernel and the
lane kernel: us per fused rollout step, same env object and buffers, interleaved rounds.

    python tools/prio_ab.py [--specs micro:65536,...] [--steps 512] [--rounds 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402
from tools.lane_ab import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--specs', default='micro:65536,tiny:65536,fives:65536,medium:65536,octa_barrage:65536,barrage:65536,micro:262144')
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
        env.rollout_steps(64)
        res = {}
        lanes = (False, True) if env.variant.cells <= 16 else (False,)
        steps = args.steps if env.variant.cells <= 36 else max(64, args.steps // 4)
        for rnd in range(args.rounds):
            for lane in lanes:
                env.set_lane_kernel(lane)
                for prio in (0, 1, 2):
                    env._L.sgx_debug_set_prio(env._h, prio)
                    env.rollout_steps(8)
                    res.setdefault((lane, prio), []).append(timed(env.rollout_steps, steps))
        env._L.sgx_debug_set_prio(env._h, 0)
        for lane in lanes:
            base = min(res[(lane, 0)])
            print("%-13s %7d games %-14s  " % (name, n, 'lane kernel' if lane else 'wave-per-game') +
                  '   '.join("prio %d: %7.2f us (%.3f)" % (p, min(res[(lane, p)]), min(res[(lane, p)]) / base) for p in (0, 1, 2)), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
