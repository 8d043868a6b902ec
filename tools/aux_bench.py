"""Launch times of the auxiliary entry points (reset, observe, standalone sampler, env info) at bench size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for version, n in (('barrage', 65536), ('standard', 65536), ('micro', 65536)):
    env = VecStrategoEnv(version, n, seed=3, auto_reset=True)
    env.reset(); env.sample_valid_actions(); env.rollout_steps(30)
    sel = (torch.arange(n, device=env.device) % 7 == 0).to(torch.uint8)
    print(version, n, "reset(all) %.0f us | reset(1/7 selected) %.0f us | observe %.0f us | sample_valid %.0f us | env_info %.0f us | step %.0f us" % (
        timed(lambda: env.reset()), timed(lambda: env.reset(env_select=sel)), timed(lambda: env.observe()),
        timed(lambda: env.sample_valid_actions()), timed(lambda: env.env_info()), timed(lambda: env.rollout_step())), flush=True)
    env.close()
