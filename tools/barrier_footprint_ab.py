"""sgx_set_steps_barrier 0 against 1 on the same ring, for rings described by games x sets: is the set COUNT or the BYTES a launch covers what
decides whether keeping a workgroup's waves in step pays?  Plain torch.empty sets, in-process A/B, us per step.
    python tools/barrier_footprint_ab.py [version=barrage] [games:sets ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, k):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    specs = ((65536, 3), (65536, 8), (65536, 12), (65536, 24), (131072, 4), (131072, 8), (262144, 2), (262144, 3), (262144, 8), (524288, 2), (524288, 4))
    if len(sys.argv) > 2:                     # games:sets[:t] ...   (:t = the slots of ONE trajectory buffer instead of separate sets)
        specs = tuple((int(a.split(':')[0]), int(a.split(':')[1])) + (('t',) if a.endswith(':t') else ()) for a in sys.argv[2:])
    for spec in specs:
        n, sets, traj = spec[0], spec[1], len(spec) > 2
        env = VecStrategoEnv(version, n, seed=3, auto_reset=True)
        env.reset()
        env.sample_valid_actions()
        gb = sets * (env.obs.numel() * 4 + env.mask.numel()) / 1e9
        free, _ = torch.cuda.mem_get_info()
        if gb * 1e9 > 0.8 * free:
            env.close()
            continue
        k = max(sets, 32)
        if traj:
            buf = env.alloc_trajectory(sets)
            run = lambda: env.rollout_trajectory(k, buf)
        else:
            env.alloc_output_ring(sets)
            run = lambda: env.rollout_steps(k, ring=True)
        run()
        res = {0: [], 1: []}
        for rnd in range(3):
            for mode in (0, 1):
                env.set_steps_barrier(mode)
                res[mode].append(timed(run, k))
        print("%s %7d games x %2d %s = %6.1f GB: drifting %s | in step %s | %+.1f %%" %
              (version, n, sets, 'slots of one buffer' if traj else 'sets', gb, ' '.join('%7.1f' % x for x in res[0]), ' '.join('%7.1f' % x for x in res[1]),
               100.0 * (min(res[1]) / min(res[0]) - 1.0)), flush=True)
        env._ring = None
        buf = run = None
        env.obs = env.mask = None
        env.close()
        del env
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
