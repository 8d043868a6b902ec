"""Diagnostic: per-phase cycle shares of the step kernel from an SGX_STAMPS build (never timed, never shipped).

Builds stratego_env_amd/_build/libstratego_mi355x_stamps.so with -DSGX_STAMPS, runs the bench workload for a few
steps and prints the median cycles between consecutive s_memtime stamps (shares, not absolute speed).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stratego_env_amd import build as B  # noqa: E402

STAMP_LIB = os.path.join(B.OUT_DIR, 'libstratego_mi355x_stamps.so')
NAMES = ['enter', 'staged', 'applied', 'mask_gen', 'results', 'mask_out', 'obs_out', 'sampled', 'writeback', 'drained']


def build():
    cmd = ['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-fvisibility=hidden', '-DSGX_STAMPS',
           '-Wno-unused-value', '-I', B.INCLUDE, B.SRC, '-o', STAMP_LIB]
    subprocess.check_call(cmd)


def main():
    import torch
    if not os.path.exists(STAMP_LIB) or os.path.getmtime(STAMP_LIB) < os.path.getmtime(B.SRC):
        build()
    B.LIB_PATH = STAMP_LIB
    from stratego_env_amd import _lib
    _lib.LIB_PATH = STAMP_LIB
    from stratego_env_amd.vec_env import VecStrategoEnv
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    env = VecStrategoEnv(version, n, seed=0x5712A7E60, auto_reset=True)
    L = _lib.load()
    L.sgx_debug_stamps.restype = C.c_void_p
    L.sgx_debug_stamps.argtypes = [C.c_void_p]
    env.reset()
    env.sample_valid_actions()
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 40        # steps played before the stamped one (how deep into the games)
    env.rollout_steps(warm - 1)
    env.rollout_step()
    torch.cuda.synchronize()
    ptr = L.sgx_debug_stamps(env._h)
    buf = torch.empty((n, 16), dtype=torch.int64, device=env.device)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n * 16 * 8), 3)
    st = buf.cpu().numpy()
    d = np.diff(st[:, :10], axis=1).astype(np.float64)
    total = (st[:, 9] - st[:, 0]).astype(np.float64)
    print("phase            median   mean    share(mean)")
    for i in range(9):
        print("%-10s->%-10s %7.0f %7.0f   %5.1f%%" % (NAMES[i], NAMES[i + 1], np.median(d[:, i]), d[:, i].mean(), 100 * d[:, i].mean() / total.mean()))
    print("wave lifetime (enter->drained): median %.0f mean %.0f cycles (s_memtime ticks)" % (np.median(total), total.mean()))
    span = st[:, 9].max() - st[:, 0].min()
    print("launch span %.0f ticks; waves x lifetime / span = %.1f concurrent waves/chip" % (span, total.sum() / span))


if __name__ == '__main__':
    main()
