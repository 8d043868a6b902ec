"""tools/ring_size_probe.py with every output set from the placement search (alloc_output_ring(tune=True)): multi-step launches into rings
of 1, 3 and 8 well-placed sets.  If a ring of 3 were helped by the Infinity Cache (a wave rewrites its slot of a set every 3 of its own
steps), a ring of 8 -- 1.5 GB of other writes between two writes of an address -- would be slower.
    python tools/ring_size_probe_tuned.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import b_min  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = 65536
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.rollout_steps(40)
    rep = env.tune_placement(max_extra_bytes=8 << 30, wide_extra_bytes=64 << 30)
    print("own set: kept %.1f us of %s" % (min(rep['obs']), ' '.join('%.0f' % t for t in rep['obs'])))
    v = env.variant
    for sets in (1, 3, 8, 3, 8):
        reps = env.alloc_output_ring(sets, tune=True, max_extra_bytes=8 << 30, wide_extra_bytes=64 << 30)
        kept = [min(r['obs']) for r in reps[1:] if r and r.get('obs')]
        row = []
        for multi in (True, False):
            env.set_multi_step(multi)
            env.rollout_steps(16, ring=True)
            best = 1e9
            for _ in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                env.rollout_steps(steps, ring=True)
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / steps * 1e3)
            byts = b_min(v, rec_bytes=env.record_bytes, fused_steps=steps if multi else 1) * n
            row.append("%s %6.1f us = %5.2f TB/s" % ('multi-step' if multi else 'per-step', best, byts / best / 1e6))
        print("ring of %d tuned sets (extra sets kept at %s us):  %s  |  %s" % (sets, ' '.join('%.0f' % k for k in kept), row[0], row[1]), flush=True)
    env.close()


if __name__ == '__main__':
    main()
