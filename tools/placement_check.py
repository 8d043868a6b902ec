"""Placement trial on this box: the library's bounded trial (sgx_alloc_outputs, <= 8 GiB extra) against
the plain first allocation and against the best of many held hipMalloc candidates (round 1's method, ~150 GB held).
Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(env, n=6):
    env.observe()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        env.observe()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    many = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
    env = VecStrategoEnv(version, n, seed=1, auto_reset=True)
    env.reset()
    t_first = timed(env)
    t0 = time.time()
    rep = env.tune_placement(max_extra_bytes=8 << 30)
    wall = time.time() - t0
    t_lib = timed(env)
    out = {"version": version, "games": n, "torch_first_us": round(t_first, 1), "library_trial_us": [round(t, 1) for t in rep['obs']],
           "library_kept_us": round(t_lib, 1), "library_trial_seconds": round(wall, 2),
           "library_peak_extra_gb": round(env.placement_peak_extra_bytes / 2.0 ** 30, 2)}
    # round 1's method for comparison: many held candidates
    keep_obs = env.obs
    cands, times = [], []
    for i in range(many):
        try:
            cands.append(torch.empty(tuple(keep_obs.shape), dtype=torch.float32, device='cuda'))
        except torch.cuda.OutOfMemoryError:
            break
        env.obs = cands[-1]
        times.append(timed(env, 4))
    env.obs = keep_obs
    out["held_candidates_us"] = [round(t, 1) for t in times]
    out["held_candidates_min_us"] = round(min(times), 1) if times else None
    out["held_candidates_gb"] = round(len(times) * keep_obs.numel() * 4 / 2.0 ** 30, 1)
    print(json.dumps(out), flush=True)
    del cands
    env.close()


if __name__ == '__main__':
    main()
