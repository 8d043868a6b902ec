"""Does the way a 64-slot trajectory buffer is MAPPED change what address translation costs?  The observation and mask tensors of the
buffer from torch.empty (hipMalloc) against HIP virtual-memory-management mappings: one physical handle or 1 GiB handles, virtual addresses aligned
to 2 MiB / 1 GiB / 4 GiB.  us per step of sgx_step_traj, 65,536 Barrage games.      python tools/vmm_traj_probe.py [slots=64]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv, _wrap_device  # noqa: E402

hip = C.CDLL('libamdhip64.so')


class Loc(C.Structure):
    _fields_ = [('type', C.c_int), ('id', C.c_int)]


class Prop(C.Structure):
    _fields_ = [('type', C.c_int), ('handle_type', C.c_int), ('location', Loc), ('win32', C.c_void_p),
                ('compression', C.c_ubyte), ('rdma', C.c_ubyte), ('usage', C.c_ushort)]


class Access(C.Structure):
    _fields_ = [('location', Loc), ('flags', C.c_int)]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d" % (what, rc))


class Vmm:
    def __init__(self, size, va_align, chunk=0, off=0):
        prop = Prop(1, 0, Loc(1, 0), None, 0, 0, 0)          # pinned, device 0
        gran = C.c_size_t()
        ck(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), 'granularity')
        g = max(gran.value, 2 << 20)
        chunk = chunk or size
        chunk = (chunk + g - 1) // g * g
        n = (size + chunk - 1) // chunk
        self.size = n * chunk
        # (the alignment argument of hipMemAddressReserve is not honoured beyond 2 MiB here: reserve more and align by hand)
        self.base, self.reserved = C.c_void_p(), self.size + va_align + (4 << 20)
        ck(hip.hipMemAddressReserve(C.byref(self.base), C.c_size_t(self.reserved), C.c_size_t(2 << 20), None, C.c_ulonglong(0)), 'reserve')
        self.ptr = C.c_void_p((self.base.value + va_align - 1) // va_align * va_align + off)
        self.reserved_ok = self.ptr.value + self.size <= self.base.value + self.reserved
        assert self.reserved_ok
        self.handles = []
        for i in range(n):
            h = C.c_void_p()
            ck(hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)), 'create')
            ck(hip.hipMemMap(C.c_void_p(self.ptr.value + i * chunk), C.c_size_t(chunk), C.c_size_t(0), h, C.c_ulonglong(0)), 'map')
            self.handles.append(h)
        acc = Access(Loc(1, 0), 3)
        ck(hip.hipMemSetAccess(self.ptr, C.c_size_t(self.size), C.byref(acc), C.c_size_t(1)), 'set access')

    def free(self):
        ck(hip.hipMemUnmap(self.ptr, C.c_size_t(self.size)), 'unmap')
        for h in self.handles:
            ck(hip.hipMemRelease(h), 'release')
        ck(hip.hipMemAddressFree(self.base, C.c_size_t(self.reserved)), 'address free')


def timed(fn, k, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / k)
    return best


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 65536
    env = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    small = env.alloc_trajectory(slots)
    obs_shape, mask_shape = tuple(small['obs'].shape), tuple(small['mask'].shape)
    obs_bytes, mask_bytes = small['obs'].numel() * 4, small['mask'].numel()
    env.rollout_trajectory(slots, small)
    print("torch.empty (hipMalloc):                                   %6.1f us per step   obs at 0x%x" % (timed(lambda: env.rollout_trajectory(slots, small), slots), small['obs'].data_ptr()), flush=True)
    rest = {k: v for k, v in small.items() if k not in ('obs', 'mask')}
    env.obs, env.mask = torch.empty_like(env.mask), torch.empty_like(env.mask)        # (drop the env's views of the buffer)
    del small
    torch.cuda.empty_cache()
    env.obs = torch.empty((n,) + obs_shape[2:], dtype=torch.float32, device=env.device)
    env.mask = torch.empty((n,) + mask_shape[2:], dtype=torch.uint8, device=env.device)
    env.observe()
    for what, align, chunk in (("one handle, addresses aligned to 2 MiB", 2 << 20, 0), ("one handle, addresses aligned to 1 GiB", 1 << 30, 0),
                               ("one handle, addresses aligned to 4 GiB", 4 << 30, 0), ("1 GiB handles, addresses aligned to 1 GiB", 1 << 30, 1 << 30),
                               ("2 GiB handles, addresses aligned to 2 GiB", 2 << 30, 2 << 30), ("256 MiB handles, addresses aligned to 256 MiB", 256 << 20, 256 << 20),
                               ("1 GiB handles, addresses aligned to 2 MiB + 2 MiB off", (1 << 30), -(1 << 30)),
                               ("32 MiB handles, addresses aligned to 2 MiB", 2 << 20, 32 << 20)):
        off = 0
        if chunk < 0:
            chunk, off = -chunk, 2 << 20
        try:
            vo, vm = Vmm(obs_bytes, align, chunk, off), Vmm(mask_bytes, align, chunk, off)
        except RuntimeError as e:
            print("%-58s %s" % (what + ':', e), flush=True)
            continue
        traj = dict(rest)
        traj['obs'] = _wrap_device(vo.ptr.value, obs_shape, torch.float32, env.device, vo)
        traj['mask'] = _wrap_device(vm.ptr.value, mask_shape, torch.uint8, env.device, vm)
        env.rollout_trajectory(slots, traj)
        us = timed(lambda: env.rollout_trajectory(slots, traj), slots)
        print("%-58s %6.1f us per step   obs at 0x%x" % (what + ':', us, vo.ptr.value), flush=True)
        env.obs = torch.empty((n,) + obs_shape[2:], dtype=torch.float32, device=env.device)
        env.mask = torch.empty((n,) + mask_shape[2:], dtype=torch.uint8, device=env.device)
        env.observe()
        del traj
        torch.cuda.synchronize()
        vo.free(); vm.free()
    env.close()


if __name__ == '__main__':
    main()
