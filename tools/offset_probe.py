"""Does step time depend on WHERE the output buffers sit?  One env, one binary; obs/mask placed at different byte
offsets inside one big allocation."""
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
print("default obs ptr %#x mask ptr %#x" % (env.obs.data_ptr(), env.mask.data_ptr()))
env.reset(); env.sample_valid_actions()
for _ in range(32): env.rollout_step()
def timeit(tag):
    ts = []
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(48): env.rollout_step()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 48 * 1e3)
    print("%-34s %s us" % (tag, " ".join("%.1f" % t for t in ts)))
timeit("default tensors")
obs_bytes, mask_bytes = n * 26800, n * 3700
big = torch.empty(obs_bytes + mask_bytes + (64 << 20), dtype=torch.uint8, device='cuda')
print("big ptr %#x" % big.data_ptr())
for off in [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, (5 << 20) + 65536 + 256, 26800 * 5, 2 << 20, 32 << 20]:
    o = big[off:off + obs_bytes].view(torch.float32).view(n, 10, 10, 67)
    m = big[off + obs_bytes:off + obs_bytes + mask_bytes].view(n, 10, 10, 37)
    env.obs, env.mask = o, m
    timeit("obs at +%d" % off)
# separate allocations again
for i in range(3):
    env.obs = torch.empty((n, 10, 10, 67), dtype=torch.float32, device='cuda')
    env.mask = torch.empty((n, 10, 10, 37), dtype=torch.uint8, device='cuda')
    timeit("fresh alloc %d obs %#x" % (i, env.obs.data_ptr()))
