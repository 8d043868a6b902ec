"""Multi-GPU sharding: games are independent, so the global env-id range is split contiguously across ranks and
every random draw is keyed by the GLOBAL id (seed, id, game, turn).  No collective is needed on the data path."""


def shard_range(total_envs: int, rank: int, world_size: int):
    """(first global env id, number of envs) of `rank`; sizes differ by at most one."""
    base, extra = divmod(int(total_envs), int(world_size))
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n
