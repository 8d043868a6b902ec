"""One library (SGX_LIB_PATH=<variant.so> SGX_ALLOW_FOREIGN_BUILD=1) on three rollout shapes of 65,536 Barrage games: a 64-slot trajectory buffer
(one plain allocation), a ring of 24 plain sets, a ring of 3 plain sets -- us per step and a checksum of everything written (equal between
variants or the variant is wrong).  usage: python tools/traj_ab.py [slots=64]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, k, reps=4):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / k)
    return best


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        flat = t.reshape(-1)
        flat = flat.view(torch.int32) if flat.element_size() == 4 else flat
        for part in flat.split(1 << 28):                              # (integer sums on the device: order-independent, exact; in pieces -- sum() widens its input)
            h.update(part.sum().cpu().numpy().tobytes())
        h.update(flat[::4097].cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 65536
    env = VecStrategoEnv('barrage', n, seed=11, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    traj = env.alloc_trajectory(slots)
    env.rollout_trajectory(slots, traj)
    d = digest([traj['obs'], traj['mask'], traj['reward'], traj['actions']])
    us = timed(lambda: env.rollout_trajectory(slots, traj), slots)
    print("%-28s %d-slot trajectory: %6.1f us per step   digest after the first call %s" % (os.path.basename(os.environ.get('SGX_LIB_PATH', 'product')), slots, us, d), flush=True)
    del traj
    env.close()
    torch.cuda.empty_cache()
    for sets in (24, 3):
        env = VecStrategoEnv('barrage', n, seed=11, auto_reset=True)
        env.reset()
        env.sample_valid_actions()
        env.alloc_output_ring(sets)
        env.rollout_steps(2 * sets + 5, ring=True)
        d = digest([t for o, m, _ in env._ring for t in (o, m)] + [env.reward, env.next_actions])
        us = timed(lambda: env.rollout_steps(96, ring=True), 96)
        print("%-28s ring of %2d plain sets: %6.1f us per step   digest %s" % ('', sets, us, d), flush=True)
        env._ring = None
        env.close()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
