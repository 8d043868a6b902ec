"""In-process A/B of builds of the library on ONE 64-slot trajectory buffer (and one ring of 24 plain sets): one env per build, same seed, all of
them writing the first one's tensors; interleaved rounds, us per step.
    SGX_ALLOW_FOREIGN_BUILD=1 python tools/traj_lib_ab.py [version:games] tools/_dev/a.so tools/_dev/b.so [...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, k):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k


def main():
    version, n = 'barrage', 65536
    if ':' in sys.argv[1]:                                   # version:games before the libraries
        version, n = sys.argv[1].split(':')[0], int(sys.argv[1].split(':')[1])
        del sys.argv[1]
    libs = sys.argv[1:]
    slots, rounds = 64, 4
    envs = [VecStrategoEnv(version, n, seed=11, auto_reset=True, lib_path=p) for p in libs]
    for e in envs:
        e.reset()
        e.sample_valid_actions()
    traj = envs[0].alloc_trajectory(slots)
    res = {}
    for rnd in range(rounds + 1):
        for i, e in enumerate(envs):
            us = timed(lambda: e.rollout_trajectory(slots, traj), slots)
            if rnd:
                res.setdefault(('64-slot trajectory', i), []).append(us)
    same = all(torch.equal(envs[0].env_info(), e.env_info()) for e in envs[1:])
    del traj
    for e in envs:
        e.obs = e.mask = None                                        # (drop the views of the buffer)
    torch.cuda.empty_cache()
    first = envs[0]
    first.obs, first.mask = torch.empty((n, first.R, first.Cc, first.p_channels), dtype=torch.float32, device=first.device), torch.empty((n, first.R, first.Cc, first.K), dtype=torch.uint8, device=first.device)
    first.observe()
    for sets in (24, 3):
        first.alloc_output_ring(sets)
        for e in envs[1:]:
            e.obs, e.mask, e.fobs = first.obs, first.mask, first.fobs
            e._ring, e._ring_owners, e._ring_pos = first._ring, first._ring_owners, 0
            e._ring_ios = (_lib.SgxStepIO * sets)()
        for rnd in range(rounds + 1):
            for i, e in enumerate(envs):
                us = timed(lambda: e.rollout_steps(96, ring=True), 96)
                if rnd:
                    res.setdefault(('ring of %d plain sets' % sets, i), []).append(us)
    same = same and all(torch.equal(envs[0].env_info(), e.env_info()) for e in envs[1:])
    print("games in the same state after the same steps: %s" % same)
    for (what, i), v in sorted(res.items()):
        print("%-24s %-24s %s  best %.1f" % (what, os.path.basename(libs[i]), ' '.join('%6.1f' % x for x in v), min(v)))
    for e in envs:
        e._ring = None
        e.close()


if __name__ == '__main__':
    main()
