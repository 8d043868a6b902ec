"""In-process A/B of the multi-step launch (steps_kernel / lane_steps_kernel: all steps of a rollout call in one launch, the games on the
chip between the steps) against one launch per step: same env object, same buffers, interleaved rounds; us per step in place, into a
ring of three output sets, with compact outputs, without outputs; and a check that both ways end in the same place.

    python tools/multi_step_ab.py [--specs barrage:65536,standard:262144,...] [--steps 128] [--rounds 2]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
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
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--specs', default='barrage:65536,standard:65536,octa_barrage:65536,medium:65536,fives:65536,standard2:32768')
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
        # same games both ways: two env objects with the same seed for the equality check, one object for the timing
        a = VecStrategoEnv(name, n, seed=77, auto_reset=True)
        b = VecStrategoEnv(name, n, seed=77, auto_reset=True)
        b.set_multi_step(False)
        a.reset(); b.reset()
        a.rollout_steps(37); b.rollout_steps(37)
        same = all(torch.equal(x, y) for x, y in ((a.obs, b.obs), (a.mask, b.mask), (a.reward, b.reward), (a.done, b.done), (a.next_actions, b.next_actions),
                                                  (a.env_info(), b.env_info())))
        sa, _ = a.export_state()
        sb, _ = b.export_state()
        same = same and torch.equal(sa, sb)
        kind = a.last_launch_kind
        b.close()
        del b, sa, sb
        env = a
        res = {}
        cases = [('in place', lambda k: env.rollout_steps(k)), ('no outputs', lambda k: env.rollout_steps(k, emit_obs=False, emit_mask=False)),
                 ('mask only', lambda k: env.rollout_steps(k, emit_obs=False))]
        for what, fn in cases:
            for rnd in range(args.rounds):
                for mode in (True, False):
                    env.set_multi_step(mode)
                    fn(8)
                    res.setdefault((what, mode), []).append(timed(fn, args.steps))
        env.alloc_output_ring(3)
        for rnd in range(args.rounds):
            for mode in (True, False):
                env.set_multi_step(mode)
                env.rollout_steps(8, ring=True)
                res.setdefault(('ring of 3', mode), []).append(timed(lambda k: env.rollout_steps(k, ring=True), args.steps))
        env.close()
        try:
            c = VecStrategoEnv(name, n, seed=77, auto_reset=True, compact_outputs=True)
            c.reset()
            c.rollout_steps(16)
            for rnd in range(args.rounds):
                for mode in (True, False):
                    c.set_multi_step(mode)
                    c.rollout_steps(8)
                    res.setdefault(('compact', mode), []).append(timed(lambda k: c.rollout_steps(k), args.steps))
            c.close()
        except Exception as e:      # noqa: BLE001
            print("  (no compact outputs on %s: %s)" % (name, e))
        print("%s %d games: multi-step launch kind %d, same results as one launch per step: %s" % (name, n, kind, same))
        for what in ('in place', 'ring of 3', 'compact', 'mask only', 'no outputs'):
            if (what, True) in res:
                m, p = min(res[(what, True)]), min(res[(what, False)])
                print("   %-11s multi-step %8.2f us   per-step launches %8.2f us   ratio %.3f   rounds %s | %s" %
                      (what, m, p, m / p, ' '.join('%.1f' % v for v in res[(what, True)]), ' '.join('%.1f' % v for v in res[(what, False)])), flush=True)
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
