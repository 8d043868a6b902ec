"""A ring of 64 placed sets (128 GB) and a 64-slot trajectory buffer written by launches of c steps each (c = 4 .. 64): do short launches --
few sets in flight per CU at a time -- avoid the translation misses of the long one (tools/ring_footprint_counters.sh)?
    python tools/ring_chunk_probe.py [sets=64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import b_min  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, k, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / k)
    return best


def main():
    most = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 65536
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.rollout_steps(40)
    env.tune_placement(max_extra_bytes=8 << 30, wide_extra_bytes=64 << 30)
    env.alloc_output_ring(most, tune=True, max_extra_bytes=8 << 30, trials=24, wide_extra_bytes=64 << 30)
    env.rollout_steps(most, ring=True)
    byts = b_min(env.variant, rec_bytes=env.record_bytes, fused_steps=64) * n
    for c in (64, 32, 16, 12, 8, 6, 4, 2):
        us = timed(lambda: [env.rollout_steps(c, ring=True) for _ in range(3 * most // c)], 3 * most)
        print("ring of %d placed sets, launches of %2d steps: %6.1f us per step = %5.2f TB/s" % (most, c, us, byts / us / 1e6), flush=True)
    env._ring = env._ring_owners = None
    env.close()
    torch.cuda.empty_cache()
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    traj = env.alloc_trajectory(most)
    env.rollout_trajectory(most, traj)

    def run(c):
        at = 0
        for _ in range(3 * most // c):
            env.rollout_trajectory(c, traj, first_slot=at)
            at = (at + c) % most
    for c in (64, 32, 16, 12, 8, 6, 4, 2):
        us = timed(lambda: run(c), 3 * most)
        print("%d-slot plain trajectory buffer, launches of %2d steps: %6.1f us per step = %5.2f TB/s" % (most, c, us, byts / us / 1e6), flush=True)
    env.close()


if __name__ == '__main__':
    main()
