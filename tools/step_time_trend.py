"""Per-chunk step time over a long rollout: separates game-phase effects from clock/thermal drift."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
n = 65536
# wake the GPU
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2.0:
    x.fill_(1.0); torch.cuda.synchronize()
env = VecStrategoEnv('barrage', n, seed=0x5712A7E60, auto_reset=True)
env.reset(); env.sample_valid_actions()
prev_games = 0
for chunk in range(24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(64):
        env.rollout_step()
    b.record(); torch.cuda.synchronize()
    info = env.env_info()
    games = int(info[:, 1].to(torch.int64).sum()); turns = float(info[:, 0].float().mean())
    nvalid = float(env.mask.view(n, -1).sum(dim=1).float().mean())
    print("steps %4d-%4d  %.1f us/step  games finished %6d  mean turn %.0f  mean valid moves %.1f" %
          (chunk * 64, chunk * 64 + 63, a.elapsed_time(b) / 64 * 1e3, games - prev_games, turns, nvalid))
    prev_games = games
