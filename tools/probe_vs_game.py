"""The multi-step rollout (ring of three) and the store-only probe on the same buffers, for rocprofv3 --pmc passes: which write-path counters differ
per byte written?    (run by tools/probe_vs_game_counters.sh)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

env = VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True)
env.reset(); env.rollout_steps(64)
env.alloc_output_ring(3)
env.rollout_steps(64, ring=True)
torch.cuda.synchronize()
for _ in range(3):
    env.rollout_steps(256, ring=True)            # steps_kernel: 256 steps x 2.0 GB
torch.cuda.synchronize()
L = env._L
t = env._ring[0][0]
us, gbs = C.c_float(), C.c_float()
for nt in (1, 8):                                # all non-temporal; every 8th sweep plain
    _lib.check(L.sgx_store_probe(0, C.c_void_p(t.data_ptr()), int(t.numel() * 4), int(t[0].numel() * 4), 64, 1, nt, 24, 0, 0, 1, 1, 2, env._stream(), C.byref(us), C.byref(gbs)), L)
    print("probe nt=%d: %.0f GB/s" % (nt, gbs.value), flush=True)
torch.cuda.synchronize()
env.close()
