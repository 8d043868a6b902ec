"""In-process A/B of the lane-per-game kernel against the wave-per-game kernel on the toy boards (sgx_set_lane_kernel): same env
object, same buffers, interleaved rounds; us per launch of a fused rollout step (sgx_step_n), outputs on and off, in place and into a
ring of three output sets, one chain and two.

    python tools/lane_ab.py [--specs micro:65536,tiny:65536,micro:262144] [--steps 512] [--rounds 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import b_min  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    fn(steps)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--specs', default='micro:65536,tiny:65536,micro:262144,tiny:262144,micro:16384')
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
        v = env.variant
        bytes_min = b_min(v, rec_bytes=env.record_bytes) * n

        def logic_only(k):
            for _ in range(k):
                env.step(env.next_actions, want_next_actions=True, emit_obs=False, emit_mask=False)

        def ring(k):
            env.rollout_steps(k, ring=True)

        def two(k):
            env.rollout_steps(k, chains=2)
        def mask_only(k):
            for _ in range(k):
                env.step(env.next_actions, want_next_actions=True, emit_obs=False, emit_mask=True)
        res = {}
        for what, fn, steps in (('in place', env.rollout_steps, args.steps), ('logic only', logic_only, min(args.steps, 128)),
                                ('mask only', mask_only, min(args.steps, 128)), ('two chains', two, args.steps)):
            for rnd in range(args.rounds):
                for mode in (True, False):
                    env.set_lane_kernel(mode)
                    fn(8)
                    res.setdefault((what, mode), []).append(timed(fn, steps))
        env.alloc_output_ring(3)
        for rnd in range(args.rounds):
            for mode in (True, False):
                env.set_lane_kernel(mode)
                ring(8)
                res.setdefault(('ring of 3', mode), []).append(timed(ring, args.steps))
        print("%s %d games (B_min %.1f MB per launch):" % (name, n, bytes_min / 1e6))
        for what in ('in place', 'ring of 3', 'two chains', 'mask only', 'logic only'):
            ln, wv = min(res[(what, True)]), min(res[(what, False)])
            print("   %-11s lane %7.2f us (frac %.3f)   wave-per-game %7.2f us (frac %.3f)   lane / wave %.3f   rounds lane %s wave %s" %
                  (what, ln, bytes_min / ln / 1e6 / 8.0, wv, bytes_min / wv / 1e6 / 8.0, ln / wv,
                   ' '.join('%.1f' % x for x in res[(what, True)]), ' '.join('%.1f' % x for x in res[(what, False)])), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
