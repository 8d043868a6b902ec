"""Three multi-step launches of `steps` steps each, 65,536 Barrage games into a ring of `sets` plain output sets: for rocprofv3 (per-dispatch
timestamps, WRITE_SIZE / FETCH_SIZE / TCC write requests).   python tools/ring8_probe.py [sets] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

sets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
env = VecStrategoEnv('barrage', 65536, seed=5, auto_reset=True)
env.reset()
env.rollout_steps(40)
env.alloc_output_ring(sets)
for _ in range(3):
    env.rollout_steps(steps, ring=True)
torch.cuda.synchronize()
env.close()
