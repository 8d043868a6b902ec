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
obs = [env.obs] + [torch.empty((n, 10, 10, 67), dtype=torch.float32, device='cuda') for _ in range(4)]
msk = [env.mask] + [torch.empty((n, 10, 10, 37), dtype=torch.uint8, device='cuda') for _ in range(4)]
def tm():
    for _ in range(6): env.rollout_step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(32): env.rollout_step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 32 * 1e3
print("obs ptrs", [hex(o.data_ptr()) for o in obs]); print("mask ptrs", [hex(m.data_ptr()) for m in msk])
print("rows = obs buffer, cols = mask buffer (us/step)")
for i, o in enumerate(obs):
    row = []
    for j, m in enumerate(msk):
        env.obs, env.mask = o, m
        row.append(tm())
    print(i, " ".join("%6.1f" % v for v in row))
# obs only (mask output disabled) and mask only
env.obs, env.mask = obs[0], msk[0]
for i, o in enumerate(obs):
    env.obs = o
    for _ in range(4): env.step(env.next_actions, want_next_actions=True, emit_mask=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(32): env.step(env.next_actions, want_next_actions=True, emit_mask=False)
    e1.record(); torch.cuda.synchronize(); print("obs-only buffer", i, "%.1f us" % (e0.elapsed_time(e1) / 32 * 1e3))
