"""Does a 64-slot trajectory buffer (one 112 GB observation tensor + 15 GB of masks) have placement classes a search could use?  The same env
writes candidates allocated one after the other, each behind a padding allocation of growing size (held while the candidate is timed).
    python tools/traj_placement_probe.py [slots=64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, k, reps=2):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / k)
    return best


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 65536
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    for pad_gb in (0, 1, 3, 7, 13, 22, 34, 0):
        pad = torch.empty(int(pad_gb * (1 << 30)), dtype=torch.uint8, device=env.device) if pad_gb else None
        traj = env.alloc_trajectory(slots)
        env.rollout_trajectory(slots, traj)
        us = timed(lambda: env.rollout_trajectory(slots, traj), slots)
        print("padding %2d GB: obs at 0x%x  %6.1f us per step" % (pad_gb, traj['obs'].data_ptr(), us), flush=True)
        o, m = torch.empty_like(env.mask), torch.empty_like(env.mask)
        env.obs, env.mask = traj['obs'][0].new_empty((n,) + tuple(traj['obs'].shape[2:])), m
        env.observe()
        del traj, pad, o
        torch.cuda.empty_cache()
    env.close()


if __name__ == '__main__':
    main()
