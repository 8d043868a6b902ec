"""Driver of tools/placement_class_counters.sh: plain torch.empty output sets are allocated (and kept) until one of the slow placement class and
one of the fast class are in hand (in-place multi-step launches, 65,536 Barrage games), then three launches of 32 steps on the FAST set and three
on the SLOW one -- the last six steps_kernel dispatches of the process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

n, steps = 65536, 32
env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
env.reset()
env.sample_valid_actions()


def run(o, m, k=steps):
    env.obs, env.mask = o, m
    env.observe()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env.rollout_steps(k); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k


sets, times = [], []
for i in range(40):
    o, m = torch.empty_like(env.obs), torch.empty_like(env.mask)
    run(o, m, 4)
    sets.append((o, m)); times.append(run(o, m))
    if len(times) >= 6 and max(times) > 1.12 * min(times):
        break
print("in place, us per step of each allocation: " + ' '.join('%.0f' % t for t in times), flush=True)
fast, slow = times.index(min(times)), times.index(max(times))
for name, i in (('fast', fast), ('slow', slow)):
    o, m = sets[i]
    print("%s set: obs at 0x%x" % (name, o.data_ptr()), flush=True)
    for _ in range(3):
        print("%s set: %.1f us per step" % (name, run(o, m)), flush=True)
env.close()
