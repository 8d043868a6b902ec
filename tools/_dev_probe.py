"""dev: a few per-step launches without outputs / with outputs from the library named by SGX_LIB_PATH (for rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
env = VecStrategoEnv('barrage', 65536, seed=77, auto_reset=True)
env.reset()
env.set_multi_step(False)
env.rollout_steps(20, emit_obs=False, emit_mask=False)
env.rollout_steps(20)
torch.cuda.synchronize()
