"""Which clocks does the driver report WHILE the multi-step rollout runs, and while the store-only probe runs?  (sysfs pp_dpm_{sclk,mclk,fclk,socclk},
the level marked '*', sampled every millisecond from a thread.)  A store-only kernel keeps the vector units idle: if power management lowers the shader
or fabric clock under it, its rate is not the memory's."""
import ctypes as C
import glob
import os
import re
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

FILES = {k: sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_' + k)) for k in ('sclk', 'mclk', 'fclk', 'socclk')}


def read_clocks():
    out = {}
    for k, fs in FILES.items():
        for f in fs:
            try:
                for line in open(f):
                    m = re.match(r'\s*\d+:\s*(\d+)\s*Mhz\s*\*', line, re.I)
                    if m:
                        out[k] = max(out.get(k, 0), int(m.group(1)))
            except OSError:
                pass
    return out


def sample_while(fn):
    stop, seen = threading.Event(), []

    def run():
        while not stop.is_set():
            seen.append(read_clocks())
            time.sleep(0.001)
    th = threading.Thread(target=run)
    th.start()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    agg = {}
    for s in seen:
        for k, v in s.items():
            agg.setdefault(k, {}).setdefault(v, 0)
            agg[k][v] += 1
    return dt, len(seen), agg


def main():
    print("clock files:", {k: len(v) for k, v in FILES.items()}, "idle:", read_clocks(), flush=True)
    env = VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True)
    env.reset(); env.rollout_steps(64)
    env.alloc_output_ring(3)
    env.rollout_steps(64, ring=True)
    torch.cuda.synchronize()
    dt, n, agg = sample_while(lambda: [env.rollout_steps(512, ring=True) for _ in range(3)])
    print("multi-step rollout, ring of 3: %.1f us per step, %d samples, clocks seen (MHz: samples): %s" % (dt / 1536 * 1e6, n, agg), flush=True)
    L = env._L
    t = env._ring[0][0]
    seg = int(t[0].numel() * 4)
    us, gbs = C.c_float(), C.c_float()

    def probe():
        _lib.check(L.sgx_store_probe(0, C.c_void_p(t.data_ptr()), int(t.numel() * 4), seg, 60, 1, 1, 24, 0, 0, 1, 1, 3, env._stream(), C.byref(us), C.byref(gbs)), L)
    dt, n, agg = sample_while(probe)
    print("store-only probe (observation-like, 24 waves per CU): %.0f GB/s, %d samples, clocks seen: %s" % (gbs.value, n, agg), flush=True)
    x = torch.empty(1 << 29, dtype=torch.float32, device='cuda')
    dt, n, agg = sample_while(lambda: [x.fill_(1.0) for _ in range(60)])
    print("torch.fill_ of 2 GiB x 60: %.0f GB/s, %d samples, clocks seen: %s" % (60 * 2.147 / dt, n, agg), flush=True)
    env.close()


if __name__ == '__main__':
    main()
