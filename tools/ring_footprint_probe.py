"""How the multi-step rollout's time per step depends on HOW MUCH memory the ring covers (round 6: a ring of 3 or 8 placed sets runs at 243 us
per step, a ring of 64 placed sets -- every one of the fast class when stepped in place -- at 288 us).  One ring of `most` separately placed sets
is allocated once; sub-rings of its first k sets (and, as a control, of its LAST k sets) are timed with the same launch (pointers in the device
table beyond 8 sets).  Then the same for k slots of ONE plain allocation (sgx_step_traj).
    python tools/ring_footprint_probe.py [most=64] [steps per launch=192]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import b_min  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
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
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    n = 65536
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.rollout_steps(40)
    env.tune_placement(max_extra_bytes=8 << 30, wide_extra_bytes=64 << 30)
    t0 = time.time()
    reps = env.alloc_output_ring(most, tune=True, max_extra_bytes=8 << 30, trials=24, wide_extra_bytes=64 << 30)
    kept = [min(r['obs']) for r in reps[1:] if r and r.get('obs')]
    print("ring of %d placed sets in %.0f s; in place: %.1f .. %.1f us" % (most, time.time() - t0, min(kept), max(kept)), flush=True)
    full, owners = list(env._ring), list(env._ring_owners)
    addrs = sorted(o.data_ptr() for o, _, _ in full)
    print("observation buffers span %.1f GB of address space" % ((addrs[-1] - addrs[0]) / 1e9 + 1.76))
    byts = b_min(env.variant, rec_bytes=env.record_bytes, fused_steps=steps) * n

    def sub_ring(sets):
        env._ring, env._ring_owners = list(sets), [None] * len(sets)
        env._ring_ios = (_lib.SgxStepIO * len(sets))()
        env._ring_pos = 0
        env.obs, env.mask, env.fobs = sets[-1]
        env.observe()
        env.rollout_steps(len(sets), ring=True)
        return timed(lambda: env.rollout_steps(steps, ring=True), steps)

    ks = [k for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96) if k <= most]
    for k in ks:
        first, last = sub_ring(full[:k]), sub_ring(full[most - k:])
        every = sub_ring(full[::most // k][:k]) if k > 1 else first
        print("ring of %2d sets (%5.1f GB): first k %6.1f us = %5.2f TB/s | last k %6.1f us | every %d-th %6.1f us" %
              (k, k * 2.0, first, byts / first / 1e6, last, most // k, every), flush=True)
    # the same sub-ring twice in the table: 2k entries over k buffers (the table's length without the footprint)
    for k in (8, 32):
        if k <= most:
            print("ring of %2d sets entered twice (%d table entries): %6.1f us" % (k, 2 * k, sub_ring(full[:k] * 2)), flush=True)
    env._ring = env._ring_owners = None
    del full, owners
    env.close()
    torch.cuda.empty_cache()
    # one plain allocation, k slots of it
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    traj = env.alloc_trajectory(most)
    env.rollout_trajectory(most, traj)
    for k in ks:
        sub = {key: t[:k] for key, t in traj.items()}
        env.rollout_trajectory(k, sub)
        us = timed(lambda: [env.rollout_trajectory(k, sub) for _ in range(max(1, steps // k))], k * max(1, steps // k))
        print("first %2d slots of one plain %d-slot allocation: %6.1f us = %5.2f TB/s" % (k, most, us, byts / us / 1e6), flush=True)
    env.close()


if __name__ == '__main__':
    main()
