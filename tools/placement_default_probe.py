"""What VecStrategoEnv(placement='search') gives a caller who changes nothing else: 65,536 Barrage games, in place, against the plain tensors of
the same process.  us per step of a rollout call (multi-step launch) and of env.step()-style launches (one per step).
    python tools/placement_default_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
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


for placement in ('plain', 'search'):
    env = VecStrategoEnv('barrage', 65536, seed=3, auto_reset=True, placement=placement)
    t0 = time.time()
    env.reset()
    torch.cuda.synchronize()
    t_reset = time.time() - t0
    env.rollout_steps(64)
    multi = timed(lambda: env.rollout_steps(128), 128)
    env.set_multi_step(False)
    per = timed(lambda: env.rollout_steps(64), 64)
    env.set_multi_step(True)
    ring_reps = env.alloc_output_ring(3)
    env.rollout_steps(64, ring=True)
    ring = timed(lambda: env.rollout_steps(128, ring=True), 128)
    rep = env.placement_report or {}
    print("placement=%-6s first reset() %.2f s; rollout call %6.1f us per step = %5.1f M steps/s; one launch per step %6.1f us = %5.1f M; ring of 3 (alloc_output_ring(3)) %6.1f us = %5.1f M; candidates %s; held for a moment %.1f GB"
          % (placement, t_reset, multi, 65536 / multi, per, 65536 / per, ring, 65536 / ring, ' '.join('%.0f' % x for x in rep.get('obs', [])) or '-', env.placement_peak_extra_bytes / 1e9), flush=True)
    env.close()
    del env
    torch.cuda.empty_cache()
