"""In-process A/B of the workgroup -> state map of the int64 state kernels (sgx_step_states / import / export): XCD ranges with the
odd / even skew (default on 8x8 ... 10x10), XCD ranges with equal shares, linear.  Same input and output tensors for every variant."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    env = VecStrategoEnv(version, n, seed=3, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    env.rollout_steps(60)
    states, players = env.export_state()
    variants = []
    ab_libs = [x for x in os.environ.get('AB_LIBS', '').split(',') if x]          # alternative builds of the library, same process
    specs = [(os.path.basename(x), {'SGX_LIB_PATH': os.path.abspath(x)}) for x in ab_libs] if ab_libs else None
    for name, envs in specs or (('xcd ranges, skew auto', {}), ('xcd ranges, equal', {'SGX_XCD_SKEW': '0'}), ('linear', {'SGX_MAP': '1'}),
                       ('xcd ranges, skew 50', {'SGX_XCD_SKEW': '50'}), ('xcd ranges, skew 150', {'SGX_XCD_SKEW': '150'})):
        for k in ('SGX_XCD_SKEW', 'SGX_MAP', 'SGX_LIB_PATH'):
            os.environ.pop(k, None)
        os.environ.update(envs)
        variants.append((name, BatchedStrategoProceduralEnv(version, n)))
    for k in ('SGX_XCD_SKEW', 'SGX_MAP', 'SGX_LIB_PATH'):
        os.environ.pop(k, None)
    m1 = variants[0][1].get_valid_moves_as_1d_mask(states, players)
    acts = torch.argmax((m1 != 0).to(torch.int8), dim=1).to(torch.int32)
    out = (torch.empty_like(states), torch.empty((n,), dtype=torch.int8, device='cuda'))
    flags = _lib.STEP_ACTIONS_1D
    ref = None
    best = {name: 1e9 for name, _ in variants}
    for rnd in range(6):
        for name, penv in variants:
            fn = lambda: penv._step_states(states, players, acts, flags, export=True, out=out)   # noqa: E731
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 10
            best[name] = min(best[name], t)
            dig = (int(out[0].sum()), int(out[1].to(torch.int64).sum()))
            if ref is None:
                ref = dig
            assert dig == ref, (name, dig, ref)
            print("round %d  %-24s %8.1f us  -> %6.1f M states/s" % (rnd, name, t * 1e6, n / t / 1e6), flush=True)
    for name, _ in variants:
        print("best  %-24s %8.1f us  -> %6.1f M states/s" % (name, best[name] * 1e6, n / best[name] / 1e6))


if __name__ == '__main__':
    main()
