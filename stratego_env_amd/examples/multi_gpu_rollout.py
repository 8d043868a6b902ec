"""Random-action self-play of G x N games on G GPUs, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port 29511 \\
        -m stratego_env_amd.examples.multi_gpu_rollout [--total-envs 2097152] [--steps 256] [--version barrage]

Games never interact, so the global env-id range is split contiguously (`sharding.shard_range`), every random draw is keyed by
the GLOBAL env id (a game's trajectory does not depend on G), and the data path has no collective: the only exchange is one
all-reduce of three counters (env steps, finished games, max-turn endings) over RCCL for the report.
"""
import argparse
import os
import time

import torch
import torch.distributed as dist

from stratego_env_amd.sharding import shard_range
from stratego_env_amd.vec_env import VecStrategoEnv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--total-envs', type=int, default=2097152)
    ap.add_argument('--steps', type=int, default=256)
    ap.add_argument('--version', default='barrage')
    ap.add_argument('--seed', type=int, default=0x5712A7E60)
    args = ap.parse_args()
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    distributed = 'RANK' in os.environ
    if distributed:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    first, n = shard_range(args.total_envs, rank, world)
    env = VecStrategoEnv(args.version, n, device=local_rank, seed=args.seed, env_id_offset=first, auto_reset=True)
    env.reset()
    env.tune_placement(wide_extra_bytes=64 << 30)          # second, wide pass if the first 8 GiB hold no fast memory
    games0 = env.env_info()[:, 1].to(torch.int64).sum()
    invalid_endings = torch.zeros((), dtype=torch.int64, device=env.device)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    chunk = 32
    for _ in range(0, args.steps, chunk):
        for _ in range(chunk):
            env.rollout_step()
            invalid_endings += env.ending_invalid.sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps_done = n * (args.steps // chunk) * chunk
    counters = torch.stack([torch.tensor(steps_done, device=env.device), env.env_info()[:, 1].to(torch.int64).sum() - games0,
                            invalid_endings])
    slowest = torch.tensor([dt], dtype=torch.float64, device=env.device)
    if distributed:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("%d %s games on %d GPU(s): %d env steps in %.2f s = %.1f M env steps/s; %d games finished, %d by max_turns" %
              (args.total_envs, args.version, world, int(counters[0]), float(slowest), int(counters[0]) / float(slowest) / 1e6,
               int(counters[1]), int(counters[2])))
    env.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
