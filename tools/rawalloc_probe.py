"""Same process, same kernel: obs/mask buffers from raw hipMalloc vs torch.empty."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratego_env_amd.vec_env import VecStrategoEnv
hip = ctypes.CDLL('libamdhip64.so')
class Raw:
    def __init__(self, nbytes):
        p = ctypes.c_void_p(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)); assert rc == 0, rc
        self.p = p.value
    def data_ptr(self): return self.p
n = 65536
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2.0:
    x.fill_(1.0); torch.cuda.synchronize()
del x
env = VecStrategoEnv('barrage', n, seed=0x5712A7E60, auto_reset=True)
env.reset()
def t_obs(label):
    env.observe()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(8): env.observe()
    e1.record(); e1.synchronize(); print("%-26s %.1f us  obs %#x" % (label, e0.elapsed_time(e1) / 8 * 1e3, env.obs.data_ptr()))
t_obs("torch default")
keep = []
for i in range(3):
    env.obs = Raw(n * 26800); env.mask = Raw(n * 3700); keep.append((env.obs, env.mask)); t_obs("raw hipMalloc %d" % i)
    env.obs = torch.empty((n, 10, 10, 67), dtype=torch.float32, device='cuda'); env.mask = torch.empty((n, 10, 10, 37), dtype=torch.uint8, device='cuda'); keep.append((env.obs, env.mask)); t_obs("torch.empty %d" % i)
