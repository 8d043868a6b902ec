import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from stratego_env_amd import _lib
from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
from stratego_env_amd.vec_env import VecStrategoEnv
n = 65536
env = VecStrategoEnv('barrage', n, seed=3, auto_reset=True); env.reset(); env.rollout_steps(60)
states, players = env.export_state()
penv = BatchedStrategoProceduralEnv('barrage', n)
m1 = penv.get_valid_moves_as_1d_mask(states, players)
acts = torch.argmax((m1 != 0).to(torch.int8), dim=1).to(torch.int32)
vec = penv._vec
new_states = torch.empty_like(states); new_players = torch.empty((n,), dtype=torch.int8, device='cuda')
io = vec._fill_io(acts, False, False, False, _lib.STEP_ACTIONS_1D); io.auto_reset = 0
def run(ch, out=True):
    _lib.check(vec._L.sgx_step_states(vec._h, states.data_ptr(), players.data_ptr(), penv.last_sanitised.data_ptr(), io,
                                      new_states.data_ptr() if out else None, new_players.data_ptr() if out else None, ch, vec._stream()), vec._L)
for ch in (1, 2, 3, 4, 1, 2):
    run(ch); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run(ch)
    torch.cuda.synchronize()
    print("chains", ch, "%.1f us" % ((time.perf_counter() - t0) / 10 * 1e6))
run(1, False); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run(1, False)
torch.cuda.synchronize(); print("no export, 1 chain %.1f us" % ((time.perf_counter() - t0) / 10 * 1e6))
