"""What precedes a burst of step launches, and how the burst's launch durations develop (run under rocprofv3 --kernel-trace and
read the series with tools/trace_series.py).  usage: burst_probe.py <prelude: idle|fill|observe|steps> [burst=60] [turn0=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

prelude = sys.argv[1] if len(sys.argv) > 1 else 'idle'
burst = int(sys.argv[2]) if len(sys.argv) > 2 else 60
turn0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
x = torch.empty(1 << 28, device='cuda')
t0 = time.time()
while time.time() - t0 < 2.0:
    x.fill_(1.0)
    torch.cuda.synchronize()
env = VecStrategoEnv('barrage', 65536, seed=0x5712A7E60, auto_reset=True)
env.reset()
env.tune_placement()
env.sample_valid_actions()
if turn0 > 1:
    env.rollout_steps(turn0 - 1)
torch.cuda.synchronize()
if prelude == 'idle':
    time.sleep(1.0)
elif prelude == 'fill':
    t0 = time.time()
    while time.time() - t0 < 0.5:
        x.fill_(1.0)
        torch.cuda.synchronize()
elif prelude == 'observe':
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(16):
            env.observe()
        torch.cuda.synchronize()
elif prelude == 'steps':                   # per-step calls, host-paced
    for _ in range(40):
        env.rollout_step()
    torch.cuda.synchronize()
env.rollout_steps(burst)
torch.cuda.synchronize()
