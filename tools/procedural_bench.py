"""Throughput of the functional operator API (BatchedStrategoProceduralEnv) on caller-provided int64 [N,34,R,C] states:
get_next_state (import -> step -> export), masks and raw observations.  States come from a short rollout."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    only = sys.argv[3].split(',') if len(sys.argv) > 3 else None      # e.g. get_next_state: a kernel trace then holds that path's launches only
    env = VecStrategoEnv(version, n, seed=3, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    env.rollout_steps(60)
    states, players = env.export_state()
    penv = BatchedStrategoProceduralEnv(version, n)
    m1 = penv.get_valid_moves_as_1d_mask(states, players)
    # first valid 1-D action of every state
    acts = torch.argmax((m1 != 0).to(torch.int8), dim=1).to(torch.int32)
    parents = penv.pack(states, players)
    children = penv.new_packed()
    mask_out = torch.empty((n, m1.shape[1]), dtype=torch.uint8, device='cuda')
    for name, fn in (('packed: expand (get_next_state)', lambda: children.expand(parents, acts)),
                     ('packed: expand + 1-D mask', lambda: children.expand(parents, acts, mask_1d_out=mask_out)),
                     ('packed: copy_from', lambda: children.copy_from(parents)),
                     ('export_state', lambda: env.export_state()),
                     ('import_state', lambda: penv._load(states, players)),
                     ('get_next_state', lambda: penv.get_next_state(states, players, acts)),
                     ('is_move_valid_by_1d_index', lambda: penv.is_move_valid_by_1d_index(states, players, acts)),
                     ('get_valid_moves_as_1d_mask', lambda: penv.get_valid_moves_as_1d_mask(states, players)),
                     ('partial obs (raw)', lambda: penv.get_partially_observable_observation_extended_channels(states, players))):
        if only and name not in only:
            continue
        t = timed(fn)
        print("%-32s %9.1f us per batch of %d  -> %8.1f M states/s" % (name, t * 1e6, n, n / t / 1e6), flush=True)


if __name__ == '__main__':
    main()
