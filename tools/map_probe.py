"""NOTE: historical experiment record -- the SGX_MAP_MODE knob this script drives was removed from the library after the
study (no game->address map changed the allocation classes); kept for the method, not runnable as is.

Placement sensitivity: several fresh output allocations x game->address map modes, one process, one binary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
n = 65536
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2.0:
    x.fill_(1.0); torch.cuda.synchronize()
del x
env = VecStrategoEnv('barrage', n, seed=0x5712A7E60, auto_reset=True)
env.reset(); env.sample_valid_actions()
for _ in range(32): env.rollout_step()
keep = []
for a in range(6):
    env.obs = torch.empty((n, 10, 10, 67), dtype=torch.float32, device='cuda')
    env.mask = torch.empty((n, 10, 10, 37), dtype=torch.uint8, device='cuda')
    keep.append((env.obs, env.mask))            # keep them alive so every round gets NEW memory
    row = []
    for mode in (0, 1, 2):
        os.environ['SGX_MAP_MODE'] = str(mode)
        for _ in range(8): env.rollout_step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(48): env.rollout_step()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 48 * 1e3)
    print("alloc %d obs %#x  chunked %.1f us  linear %.1f us  chunk64 %.1f us" % (a, env.obs.data_ptr(), *row))
