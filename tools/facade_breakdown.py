"""Where an N = 1 env.step() spends its time: the bare library call (launch + kernel + wait) against the whole facade step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratego_env_amd import GameVersions, ObservationModes, _lib  # noqa: E402
from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv  # noqa: E402

env = StrategoMultiAgentEnv({'version': GameVersions.BARRAGE, 'human_inits': True, 'observation_mode': ObservationModes.PARTIALLY_OBSERVABLE})
np.random.seed(1)
obs = env.reset()
vec = env._vec
io = env._step_io
io.flags = 0
env._act_word[0] = -1            # an invalid action: the state stays, every output is still rendered
st = vec._stream()
for n in (2000, 2000):
    t0 = time.perf_counter()
    for _ in range(n):
        vec._L.sgx_step_sync(vec._h, io, st)
    dt = time.perf_counter() - t0
    print("bare sgx_step_sync: %.1f us per call" % (dt / n * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    vec._L.sgx_step(vec._h, io, st)
import torch
torch.cuda.synchronize()
print("sgx_step enqueue only: %.1f us per call" % ((time.perf_counter() - t0) / 2000 * 1e6))
hv = env._hview
t0 = time.perf_counter()
for _ in range(2000):
    d = env._obs_dict(hv['obs'][0], None, hv['mask'][0], 1)
print("_obs_dict (host copies): %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    valid = np.flatnonzero(d['valid_actions_mask'].reshape(-1)); a = int(valid[np.random.randint(len(valid))])
print("caller's action choice: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
env.close()
