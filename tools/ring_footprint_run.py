"""Driver of tools/ring_footprint_counters.sh: one ring of 64 placed sets; launches of 64 steps into its first 8 sets (16 GB), then into all
64 (128 GB).  Dispatch order of steps_kernel: [touch 8] [8] [8] [touch 64] [64] [64]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

n, most, steps = 65536, 64, 64
env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
env.reset()
env.set_multi_step(False)
env.rollout_steps(40)
env.tune_placement(max_extra_bytes=8 << 30, wide_extra_bytes=64 << 30)
env.alloc_output_ring(most, tune=True, max_extra_bytes=8 << 30, trials=24, wide_extra_bytes=64 << 30)
env.set_multi_step(True)
full, owners = list(env._ring), list(env._ring_owners)
for k in (8, most):
    env._ring = full[:k]
    env._ring_ios = (_lib.SgxStepIO * k)()
    env._ring_pos = 0
    env.obs, env.mask, env.fobs = full[k - 1]
    env.observe()
    for i in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.rollout_steps(steps, ring=True); b.record()
        torch.cuda.synchronize()
        print("ring of %2d sets, launch %d: %.1f us per step" % (k, i, a.elapsed_time(b) * 1e3 / steps), flush=True)
env._ring = full
env.close()
