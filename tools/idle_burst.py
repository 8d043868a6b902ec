"""Mean launch time of a burst of 25 step launches (sgx_step_n) as a function of the idle time before it, same env and buffers;
each burst is preceded by 0.1 s of state-preserving observe launches (the GPU at its steady load), a synchronisation and the idle time."""
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
env.sample_valid_actions()
env.rollout_steps(100)
torch.cuda.synchronize()
burst = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for rnd in range(3):
    for idle_ms in (0, 0.3, 1, 3, 10, 50, 300):
        t0 = time.time()
        while time.time() - t0 < 0.1:
            for _ in range(8):
                env.observe()
            torch.cuda.synchronize()
        time.sleep(idle_ms / 1e3)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env.rollout_steps(burst)
        b.record()
        torch.cuda.synchronize()
        print("round %d  idle %6.1f ms: %6.1f us per launch" % (rnd, idle_ms, a.elapsed_time(b) / burst * 1e3), flush=True)
