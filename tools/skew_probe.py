"""Experiment record: alternative workgroup->game maps (builds with -DSGX_XCD_SKEW, since removed from the
kernel) timed on torch allocations and on hipExtMallocWithFlags(contiguous / default) buffers, all libraries on the same buffer."""
import ctypes as C, os, sys, time, glob
sys.path.insert(0, '/root/repo')
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
hip = C.CDLL('libamdhip64.so')
class Raw:
    def __init__(self, p): self.p = p
    def data_ptr(self): return self.p
def timed(fn, n=6):
    fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2: x.fill_(1.0); torch.cuda.synchronize()
del x
libs = sorted(glob.glob('/root/repo/stratego_env_amd/_build/var_*.so'))
envs = [VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True, lib_path=l) for l in libs]
for e in envs: e.reset()
nbytes = envs[0].obs.numel() * 4
bufs = [('torch#%d' % i, torch.empty_like(envs[0].obs)) for i in range(6)]
for flag, name in ((4, 'contiguous'), (4, 'contiguous'), (0, 'ext default'), (0, 'ext default')):
    p = C.c_void_p(); assert hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flag)) == 0
    bufs.append((name, Raw(p.value)))
print("%-14s %s" % ("buffer", " ".join("%12s" % os.path.basename(l)[4:-3] for l in libs)))
for name, b in bufs:
    row = []
    for e in envs:
        e.obs = b
        row.append(timed(e.observe))
    print("%-14s %s" % (name, " ".join("%12.1f" % t for t in row)), flush=True)
