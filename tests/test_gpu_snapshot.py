"""VecStrategoEnv.snapshot() / restore(): the packed records of every game copied into a pool without output tensors (sgx_copy_envs) and
back -- boards, counters and game numbers, i.e. everything the rules and the counter RNG depend on.  bench.py's solo anchors rely on it:
rank 0 plays a leg's steps alone, puts the games back, and the side-by-side run is that of a run without the anchor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,n', [('barrage', 700), ('micro', 1500), ('standard2', 30)])
def test_restore_replays_the_same_trajectory(name, n):
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv(name, n, seed=31337, env_id_offset=12, auto_reset=True)
    env.reset()
    env.rollout_steps(17)                                           # somewhere mid-game, some envs already in their second game
    snap = env.snapshot()
    assert snap.obs is None and snap.mask is None and snap.num_envs == n
    st0, pl0 = env.export_state()
    info0 = env.env_info().clone()
    env.rollout_steps(40)
    first = [t.clone() for t in (env.obs, env.mask, env.reward, env.done, env.player, env.next_actions, env.env_info())]
    assert not torch.equal(env.env_info(), info0)
    env.restore(snap)
    st1, pl1 = env.export_state()
    assert torch.equal(st0, st1) and torch.equal(pl0, pl1) and torch.equal(env.env_info(), info0)      # game numbers included
    env.rollout_steps(40)
    again = [env.obs, env.mask, env.reward, env.done, env.player, env.next_actions, env.env_info()]
    for a, b in zip(first, again):
        assert torch.equal(a, b)
    # the snapshot itself is untouched and can be used again, also into a ring of output sets
    env.alloc_output_ring(3)
    env.restore(snap)
    env.rollout_steps(40, ring=True)
    for a, b in zip(first, [env.obs, env.mask, env.reward, env.done, env.player, env.next_actions, env.env_info()]):
        assert torch.equal(a, b)
    snap.close()
    env.close()


def test_pools_without_outputs_and_ring_misuse():
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv('tiny', 64, seed=1, auto_reset=True)
    env.reset()
    with pytest.raises(ValueError, match='alloc_output_ring'):
        env.rollout_steps(3, ring=True)                             # (used to be an AttributeError)
    pool = VecStrategoEnv('tiny', 64, outputs=False)
    with pytest.raises(ValueError):
        pool.alloc_output_ring(2)
    with pytest.raises(ValueError):
        VecStrategoEnv('tiny', 8, outputs=False, full_obs=True)
    with pytest.raises(ValueError):
        env.restore(VecStrategoEnv('tiny', 32, outputs=False))
    pool.close()
    env.close()
