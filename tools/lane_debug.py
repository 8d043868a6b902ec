import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = VecStrategoEnv('micro', n, seed=1, auto_reset=True)
env.set_lane_kernel(which.startswith('lane'))
if which.endswith('mask'):
    env._L.sgx_reset(env._h, None, None, None, None); env.observe(emit_obs=False); torch.cuda.synchronize(); print(which, 'mask ok')
elif which.endswith('obs'):
    env._L.sgx_reset(env._h, None, None, None, None); env.observe(emit_mask=False); torch.cuda.synchronize(); print(which, 'obs ok')
elif which.endswith('none'):
    env._L.sgx_reset(env._h, None, None, None, None); env.observe(emit_mask=False, emit_obs=False); torch.cuda.synchronize(); print(which, 'none ok')
else:
    env.reset(); torch.cuda.synchronize(); print(which, 'reset ok')
