"""is the Micro ring's DRAM-side rate a matter of where its sets live?  Ring sets carved out of ONE large buffer that the placement search
picked at a DRAM-side size (a 524,288-game Micro env's outputs: 1.7 GB) against plain torch.empty sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd import _lib
from stratego_env_amd.vec_env import VecStrategoEnv

def timed(fn, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(); fn(steps); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3

name, n, big = sys.argv[1] if len(sys.argv) > 1 else 'micro', 65536, 8
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2: x.fill_(1.0); torch.cuda.synchronize()
del x
env = VecStrategoEnv(name, n, seed=3, auto_reset=True)
env.reset(); env.rollout_steps(30)
env.alloc_output_ring(3)
plain = [timed(lambda k: env.rollout_steps(k, ring=True), 256) for _ in range(3)]
inplace = [timed(lambda k: env.rollout_steps(k), 256) for _ in range(3)]
# the large searched buffer
helper = VecStrategoEnv(name, n * big, seed=4, auto_reset=True)
helper.reset()
rep = helper.tune_placement(max_extra_bytes=8 << 30, wide_extra_bytes=32 << 30)
print("helper (%d games) search: kept %.1f us of %s" % (n * big, min(rep['obs']), ' '.join('%.0f' % t for t in rep['obs'])))
obs_big, mask_big = helper.obs, helper.mask
R, Cc, K = env.R, env.Cc, env.K
sets = []
for i in range(3):
    o = obs_big.view(-1)[i * n * R * Cc * 67:(i + 1) * n * R * Cc * 67].view(n, R, Cc, 67)
    m = mask_big.view(-1)[i * n * R * Cc * K:(i + 1) * n * R * Cc * K].view(n, R, Cc, K)
    sets.append((o, m, None))
env._ring = sets
env._ring_owners = [helper, helper, helper]
env._ring_pos = 0
env.obs, env.mask = sets[0][0], sets[0][1]
env.rollout_steps(8, ring=True)
carved = [timed(lambda k: env.rollout_steps(k, ring=True), 256) for _ in range(3)]
carved_inplace = [timed(lambda k: env.rollout_steps(k), 256) for _ in range(3)]
print("%s %d games, us per step: ring of 3 plain %s | in place plain %s | ring of 3 carved from the searched buffer %s | in place in the carved set %s" %
      (name, n, ' '.join('%.1f' % v for v in plain), ' '.join('%.1f' % v for v in inplace), ' '.join('%.1f' % v for v in carved), ' '.join('%.1f' % v for v in carved_inplace)))
