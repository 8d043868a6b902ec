import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
n = 65536
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 1.5:
    x.fill_(1.0); torch.cuda.synchronize()
del x
mode = sys.argv[1]
if mode == 'slab_first':
    big = torch.empty(n * 30500 + (4 << 20), dtype=torch.uint8, device='cuda')
env = VecStrategoEnv('barrage', n, seed=0x5712A7E60, auto_reset=True)
if mode == 'slab_after':
    big = torch.empty(n * 30500 + (4 << 20), dtype=torch.uint8, device='cuda')
if mode != 'default':
    env.obs = big[:n * 26800].view(torch.float32).view(n, 10, 10, 67)
    env.mask = big[n * 26800:n * 30500].view(n, 10, 10, 37)
env.reset(); env.sample_valid_actions()
for _ in range(32): env.rollout_step()
ts = []
for r in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(48): env.rollout_step()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 48 * 1e3)
print("%-11s obs %#x mask %#x  %s us" % (mode, env.obs.data_ptr(), env.mask.data_ptr(), " ".join("%.1f" % t for t in ts)))
