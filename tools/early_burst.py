"""Bursts of 20 step launches at early turns (all games in lockstep from reset) against bursts deep into the games, same env and
buffers, each preceded by 0.1 s of observe launches and no idle time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

env = VecStrategoEnv('barrage', 65536, seed=0x5712A7E60, auto_reset=True)
env.reset()
rep = env.tune_placement()
print("placement: first %.1f kept %.1f" % (rep['obs'][0], min(rep['obs'])))


def settle():
    t0 = time.time()
    while time.time() - t0 < 0.1:
        for _ in range(8):
            env.observe()
        torch.cuda.synchronize()


def burst(k):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.rollout_steps(k)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


for rnd in range(3):
    env.reset()
    env.sample_valid_actions()
    turn = 0
    line = []
    for k in (5, 20, 20, 20, 40, 100, 20, 200, 20):
        settle()
        line.append("turns %d-%d: %.1f" % (turn + 1, turn + k, burst(k)))
        turn += k
    print("round %d  " % rnd + "  ".join(line), flush=True)


def counters():
    return int(env.env_info()[:, 1].to(torch.int64).sum())


def fill_wake(seconds):
    x = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
    t0 = time.time()
    while time.time() - t0 < seconds:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x


print("-- what stands between the settle phase and the burst (turns 6-25 each time)")
for rnd in range(2):
    for name in ('settle,burst', 'settle,5 steps,burst', 'settle,5 steps,counters,2 syncs,burst', 'fill 0.3 s,settle,sample,5 steps,counters,burst',
                 'settle,5 steps,counters,burst(20) timed by perf_counter too', 'settle,5 steps,sleep 20 ms,burst'):
        env.reset()
        if name.startswith('fill'):
            fill_wake(0.3)
        settle()
        env.sample_valid_actions()
        if '5 steps' in name:
            for _ in range(5):
                env.rollout_step()
        else:
            env.rollout_steps(5)
        if 'counters' in name:
            counters()
            torch.cuda.synchronize()
            torch.cuda.synchronize()
        if 'sleep' in name:
            torch.cuda.synchronize()
            time.sleep(0.02)
        t0 = time.perf_counter()
        us = burst(20)
        wall = (time.perf_counter() - t0) / 20 * 1e6
        print("round %d  %-62s %.1f us per launch (wall %.1f)" % (rnd, name, us, wall), flush=True)
