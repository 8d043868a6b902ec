"""Micro / Tiny ring of three vs in place under forced NT / plain stores, multi-step lane kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
def timed(fn, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(); fn(steps); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2: x.fill_(1.0); torch.cuda.synchronize()
del x
for name in ('micro', 'tiny'):
    for sets in (1, 3, 8):
        env = VecStrategoEnv(name, 65536, seed=3, auto_reset=True)
        env.reset(); env.rollout_steps(30)
        if sets > 1: env.alloc_output_ring(sets)
        fn = (lambda k: env.rollout_steps(k, ring=True)) if sets > 1 else (lambda k: env.rollout_steps(k))
        out = []
        for mode, label in ((None, 'auto'), (True, 'nt'), (False, 'plain')):
            env.set_nt_stores(mode)
            fn(8)
            out.append('%s %s' % (label, ' '.join('%.1f' % timed(fn, 256) for _ in range(2))))
        print('%s ring of %d: %s' % (name, sets, ' | '.join(out)), flush=True)
        env.close()
