"""Does the allocation FLAG change the speed class of the obs buffer?  hipMalloc vs
hipExtMallocWithFlags(default / fine-grained / uncached / contiguous): time sgx_observe writing into each."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

hip = C.CDLL('libamdhip64.so')


class Raw:
    """Just enough of a tensor for VecStrategoEnv.observe()."""

    def __init__(self, ptr):
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr


def timed(fn, n=6):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    env = VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True)
    env.reset()
    nbytes = env.obs.numel() * 4
    keep = []
    print("torch.empty   %s" % " ".join("%6.1f" % t for t in [timed(env.observe)] + [
        (setattr(env, 'obs', torch.empty_like(env.obs)), timed(env.observe))[1] for _ in range(5)]))
    for name, flag in (('hipMalloc', None), ('ext default', 0), ('ext finegrained', 1), ('ext uncached', 3), ('ext contiguous', 4)):
        ts = []
        for _ in range(6):
            p = C.c_void_p()
            rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) if flag is None else \
                hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flag))
            if rc != 0:
                ts.append(float('nan'))
                continue
            keep.append(p)
            env.obs = Raw(p.value)
            ts.append(timed(env.observe))
        print("%-15s %s" % (name, " ".join("%6.1f" % t for t in ts)), flush=True)


if __name__ == '__main__':
    main()
