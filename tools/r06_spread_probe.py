"""Round 6, the one placement lead left (time-boxed): plain torch.empty output sets run at 6.4-6.7 TB/s as a ring of 1-3 sets and at 8.1 TB/s as
a ring of 6-8 (profiles/r05_ring_size_probe.log).  Is it the NUMBER of allocations a launch's stores are spread over, or the amount of
address space?  Multi-step launches of 65,536 Barrage games into
  A  n separate plain allocations (the r05 probe),
  B  n sets carved back to back out of ONE plain allocation,
  C  n sets carved out of one plain allocation at a stride of 8 GiB,
  T  a trajectory buffer [T, N, ...] in ONE plain allocation (sgx_step_traj, n_steps = T),
us per step (best of 3 calls of 96 steps) and B_min bytes / time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

N = 65536
BMIN = 30512.0


def timed(fn, k):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / k)
    return best


def ring_of(env, tensors):
    """install a ring of (obs, mask) pairs"""
    env._ring = [(o, m, None) for o, m in tensors]
    env._ring_owners = [None] * len(tensors)
    env._ring_ios = (_lib.SgxStepIO * len(tensors))()
    env._ring_pos = 0
    env.obs, env.mask = tensors[0]
    env.observe()


def report(tag, us):
    print("%-64s %7.1f us per step = %5.2f TB/s" % (tag, us, BMIN * N / us / 1e6), flush=True)


def main():
    dev = torch.device('cuda', 0)
    env = VecStrategoEnv('barrage', N, seed=3, auto_reset=True)
    env.reset(); env.rollout_steps(40)
    oshape, mshape = tuple(env.obs.shape), tuple(env.mask.shape)
    on, mn = env.obs.numel(), env.mask.numel()
    K = 96
    for n in (1, 3, 8):
        sets = [(torch.empty(oshape, dtype=torch.float32, device=dev), torch.empty(mshape, dtype=torch.uint8, device=dev)) for _ in range(n)]
        ring_of(env, sets)
        report("A  ring of %d separate plain allocations" % n, timed(lambda: env.rollout_steps(K, ring=True), K))
        assert env.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
        del sets
        big_o = torch.empty((n * on,), dtype=torch.float32, device=dev)
        big_m = torch.empty((n * mn,), dtype=torch.uint8, device=dev)
        sets = [(big_o[i * on:(i + 1) * on].view(oshape), big_m[i * mn:(i + 1) * mn].view(mshape)) for i in range(n)]
        ring_of(env, sets)
        report("B  ring of %d sets back to back in ONE plain allocation" % n, timed(lambda: env.rollout_steps(K, ring=True), K))
        del sets, big_o, big_m
        if n > 1:
            stride = (8 << 30) // 4
            big_o = torch.empty(((n - 1) * stride + on,), dtype=torch.float32, device=dev)
            big_m = torch.empty((n * mn,), dtype=torch.uint8, device=dev)
            sets = [(big_o[i * stride:i * stride + on].view(oshape), big_m[i * mn:(i + 1) * mn].view(mshape)) for i in range(n)]
            ring_of(env, sets)
            report("C  ring of %d sets at 8 GiB strides in one plain allocation (%d GiB)" % (n, (n - 1) * 8 + 2), timed(lambda: env.rollout_steps(K, ring=True), K))
            del sets, big_o, big_m
        torch.cuda.empty_cache()
    env._ring = None
    env.obs = torch.empty(oshape, dtype=torch.float32, device=dev); env.mask = torch.empty(mshape, dtype=torch.uint8, device=dev)
    env.observe()
    for T in (4, 8, 16, 32, 64):
        traj = env.alloc_trajectory(T)
        env.rollout_trajectory(T, traj)
        reps = max(1, 96 // T)
        us = timed(lambda: [env.rollout_trajectory(T, traj) for _ in range(reps)], reps * T)
        assert env.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
        report("T  trajectory buffer of %2d slots in one plain allocation (%5.1f GB)" % (T, T * (on * 4 + mn) / 1e9), us)
        env.obs, env.mask = torch.empty(oshape, dtype=torch.float32, device=dev), torch.empty(mshape, dtype=torch.uint8, device=dev)
        env.reward = torch.zeros((N, 2), dtype=torch.float32, device=dev); env.done = torch.zeros((N,), dtype=torch.uint8, device=dev)
        env.player = torch.ones((N,), dtype=torch.int8, device=dev); env.invalid_action = torch.zeros((N,), dtype=torch.uint8, device=dev)
        env.ending_invalid = torch.zeros((N,), dtype=torch.uint8, device=dev)
        env.observe()
        del traj
        torch.cuda.empty_cache()
    env.close()


if __name__ == '__main__':
    main()
