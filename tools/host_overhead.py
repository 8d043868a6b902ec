"""Host cost of one step call: Python `rollout_step()` vs the C loop `sgx_step_n`, on a batch small enough to be launch-bound."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
env = VecStrategoEnv('micro', 256, seed=1, auto_reset=True)
env.reset(); env.sample_valid_actions()
for _ in range(200): env.rollout_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5000
for _ in range(n): env.rollout_step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("python rollout_step: %.1f us per call issue, %.1f us incl. drain" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
t0 = time.perf_counter()
env.rollout_steps(n)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("sgx_step_n: %.1f us per step issue, %.1f us incl. drain" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
