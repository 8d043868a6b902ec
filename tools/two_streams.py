import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
ver = sys.argv[1] if len(sys.argv) > 1 else 'micro'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2.0:
    x.fill_(1.0); torch.cuda.synchronize()
def bench(parts, steps=256):
    n = N // parts
    envs = [VecStrategoEnv(ver, n, seed=1, env_id_offset=i * n, auto_reset=True) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e, s in zip(envs, streams):
        with torch.cuda.stream(s):
            e.reset(); e.sample_valid_actions(); e.rollout_steps(32)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(5):
        t0 = time.perf_counter()
        for e, s in zip(envs, streams):
            with torch.cuda.stream(s):
                e.rollout_steps(steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    for e in envs: e.close()
    return best
for parts in (1, 2, 4, 8):
    us = bench(parts)
    print("%s %d games as %d handle(s) on %d stream(s): %.1f us per step of all games -> %.1f M steps/s" % (ver, N, parts, parts, us, N / us))
