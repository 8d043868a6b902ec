"""The lane-per-game kernel (stratego_env_amd/csrc/sgx_lane_kernel.h: boards of at most 16 cells, one game per lane) on the GPU:
every output of every step against the CPU oracle -- with the kernel forced, so that the test cannot pass on the wave-per-game
kernel -- and, output by output, against the wave-per-game kernel on the same inputs."""
import numpy as np
import pytest

from stratego_env_amd import config
from stratego_env_amd.config import custom_variant

pytestmark = pytest.mark.gpu

#                                                              spy scout miner sgt lt cpt maj col gen mar flag bomb
CUSTOM = {
    # every rule on 16 cells: scouts, spy / marshal, miner / bomb, a lake
    'zoo44': custom_variant(4, 4, max_turns=60, obstacle_locations=((1, 1),), piece_counts=(1, 2, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1),
                            initial_state_usable_rows=2, name='zoo44'),
    # three sergeants: captured counts normalise to thirds -- values without a 4-bit code (the patched floats of the lane kernel)
    'thirds44': custom_variant(4, 4, max_turns=80, piece_counts=(0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 1, 0), initial_state_usable_rows=1, name='thirds44'),
    'tall43': custom_variant(4, 3, max_turns=50, piece_counts=(0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1), initial_state_usable_rows=2, name='tall43'),
}


@pytest.fixture(autouse=True)
def _custom_names():
    config.VARIANTS.update(CUSTOM)
    yield
    for k in CUSTOM:
        config.VARIANTS.pop(k, None)


@pytest.mark.parametrize('name,n_envs,n_steps', [('micro', 64, 150), ('micro', 200, 60), ('tiny', 64, 250), ('tiny', 130, 120),
                                                 ('zoo44', 96, 250), ('thirds44', 96, 300), ('tall43', 70, 200)])
def test_lane_kernel_step_bit_exact_vs_oracle(name, n_envs, n_steps):
    """tests/test_gpu_parity.py's step-by-step comparison (mask, observation, rewards, flags, sampler, int64 state; 10 % garbage
    actions; auto-reset) without terminal-observation buffers, i.e. on the lane kernel."""
    from tests.test_gpu_parity import test_step_bit_exact_vs_oracle
    test_step_bit_exact_vs_oracle(name, n_envs, n_steps, 0.1, seed_salt=3, final_obs=False, lane_kernel=True)


@pytest.mark.parametrize('name,n', [('micro', 1), ('micro', 63), ('micro', 65), ('micro', 4097), ('tiny', 1000), ('zoo44', 777), ('thirds44', 513)])
def test_lane_and_wave_kernels_agree(name, n):
    """Same seed, same actions (garbage included), one env on each kernel: identical outputs and int64 states after every step,
    for ragged batch sizes (partial waves, partial sub-batches)."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=4242, env_id_offset=17, auto_reset=True)
    b = VecStrategoEnv(name, n, seed=4242, env_id_offset=17, auto_reset=True)
    a.set_lane_kernel(True)
    b.set_lane_kernel(False)
    a.reset(); b.reset()
    rs = np.random.RandomState(5)
    NA = a.variant.num_spatial_actions
    for t in range(70):
        for x, y in ((a.obs, b.obs), (a.mask, b.mask), (a.player, b.player)):
            assert torch.equal(x, y), (name, n, t)
        acts = a.sample_valid_actions().clone()
        assert torch.equal(acts, b.sample_valid_actions())
        bad = torch.from_numpy(rs.rand(n) < 0.08).to(acts.device)
        acts = torch.where(bad, torch.from_numpy(rs.randint(-2, NA + 2, size=n).astype(np.int32)).to(acts.device), acts)
        a.step(acts, want_next_actions=True)
        b.step(acts, want_next_actions=True)
        for x, y in ((a.reward, b.reward), (a.done, b.done), (a.invalid_action, b.invalid_action), (a.ending_invalid, b.ending_invalid),
                     (a.next_actions, b.next_actions)):
            assert torch.equal(x, y), (name, n, t)
        if t % 10 == 9:
            sa, pa = a.export_state()
            sb, pb = b.export_state()
            assert torch.equal(sa, sb) and torch.equal(pa, pb), (name, n, t)
    assert int(a.done.sum()) >= 0 and int(a.invalid_action.sum()) == int(b.invalid_action.sum())
    # the raw (un-normalised) observation and a forced non-temporal sweep
    a.observe(raw=True); b.observe(raw=True)
    assert torch.equal(a.obs, b.obs)
    a.set_nt_stores(True)
    a.observe(); b.observe()
    assert torch.equal(a.obs, b.obs) and torch.equal(a.mask, b.mask)
    a.close(); b.close()


def test_lane_kernel_rollout_digests_at_full_size():
    """65,536 + 17 Micro games, 64 fused steps on the lane kernel (sgx_step_n), then sampled games against the oracle's digests of the
    last step and its counters (so_rollout_ex)."""
    import torch
    from oracle import oracle as orc
    from stratego_env_amd.vec_env import VecStrategoEnv
    from tests.helpers import oracle_cvariant
    N, T, seed = 65536 + 17, 64, 0xBEEF
    for name in ('micro', 'tiny'):
        env = VecStrategoEnv(name, N, seed=seed, auto_reset=True)
        env.set_lane_kernel(True)
        env.reset()
        env.rollout_steps(T)
        torch.cuda.synchronize()
        assert int(env.invalid_action.sum()) == 0
        cv = oracle_cvariant(name)
        ids = [0, 1, 63, 64, 65, 4095, 40000, 65535, 65536, N - 1]
        idx = torch.tensor(ids, device=env.device)
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
        rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
        ei, info = env.ending_invalid[idx].cpu().numpy(), env.env_info()[idx].cpu().numpy()
        for i, e in enumerate(ids):
            r = orc.rollout_ex(cv, seed, e, 1, T, threads=1)
            assert int(r['last_digests'][0]) == orc.step_digest(mk[i], ob[i], rw[i], dn[i], pl[i], ei[i]), (name, e)
            assert np.array_equal(r['info'][0], info[i]), (name, e)
        env.close()


@pytest.mark.parametrize('name,n,T', [('micro', 64, 2), ('micro', 200, 33), ('micro', 4097, 64), ('tiny', 130, 41), ('zoo44', 777, 50),
                                      ('thirds44', 513, 120), ('tall43', 70, 64)])
def test_multi_step_launch_equals_step_by_step(name, n, T):
    """sgx_step_n / sgx_step_ring on boards of at most 16 cells run all steps of a call in ONE launch (lane_steps_kernel: the games in the
    registers of one wave per workgroup, three more waves emitting the observations of the step before): every output of the last step,
    every output set of a ring (the last three steps), the int64 states, counters and the next draw equal those of one wave-per-game
    launch per step -- ragged batches (partial workgroups, partial sub-batches), piece sets with uncoded thirds, a first action that is
    garbage."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=777, env_id_offset=9, auto_reset=True)          # default: the multi-step launch where eligible
    b = VecStrategoEnv(name, n, seed=777, env_id_offset=9, auto_reset=True)
    b.set_lane_kernel(False)                                                          # one wave-per-game launch per step
    a.reset(); b.reset()
    a.sample_valid_actions(); b.sample_valid_actions()
    # some envs start from a garbage action: not applied, the next one is drawn afresh
    bad = torch.arange(n, device=a.device) % 7 == 3
    for e in (a, b):
        e.next_actions[bad] = -5
        e._next_actions_fresh = True
    from stratego_env_amd import _lib
    a.rollout_steps(T)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP                              # (the test must not pass on another kernel)
    for _ in range(T):
        b.rollout_steps(1)
    assert b.last_launch_kind == _lib.LAUNCH_WAVE
    for x, y in ((a.obs, b.obs), (a.mask, b.mask), (a.reward, b.reward), (a.done, b.done), (a.player, b.player), (a.next_actions, b.next_actions),
                 (a.ending_invalid, b.ending_invalid), (a.env_info(), b.env_info())):
        assert torch.equal(x, y), (name, n, T)
    sa, pa = a.export_state()
    sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    # a ring of three output sets: the last three steps' observations and masks, set by set; then in place again, and one more single step
    ra = a.alloc_output_ring(3)
    rb = b.alloc_output_ring(3)
    del ra, rb
    a.rollout_steps(T + 1, ring=True)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP
    b.rollout_steps(T + 1, ring=True)
    for (oa, ma, _), (ob, mb, _) in zip(a._ring, b._ring):
        assert torch.equal(oa, ob) and torch.equal(ma, mb), (name, n, T, 'ring')
    assert torch.equal(a.obs, b.obs) and torch.equal(a.env_info(), b.env_info()) and torch.equal(a.next_actions, b.next_actions)
    a.set_multi_step(False)                                                          # the same env object, one launch per step again
    a.rollout_steps(5); b.rollout_steps(5)
    assert a.last_launch_kind == _lib.LAUNCH_WAVE
    a.set_multi_step(True)
    a.rollout_steps(3); b.rollout_steps(3)
    a.rollout_step(); b.rollout_step()
    for x, y in ((a.obs, b.obs), (a.mask, b.mask), (a.reward, b.reward), (a.done, b.done), (a.next_actions, b.next_actions), (a.env_info(), b.env_info())):
        assert torch.equal(x, y)
    assert int(a.invalid_action.sum()) == int(b.invalid_action.sum())
    a.close(); b.close()


def test_multi_step_launch_with_forced_non_temporal_stores_and_raw_players():
    """The multi-step launch under sgx_set_nt_stores(1) and without auto-reset (finished games stay finished: their no-op-only masks and
    terminal observations repeat) equals the per-step launches."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    for auto in (True, False):
        a = VecStrategoEnv('micro', 1000, seed=5, auto_reset=auto)
        b = VecStrategoEnv('micro', 1000, seed=5, auto_reset=auto)
        a.set_nt_stores(True)
        b.set_multi_step(False)
        a.reset(); b.reset()
        a.rollout_steps(30); b.rollout_steps(30)
        for x, y in ((a.obs, b.obs), (a.mask, b.mask), (a.reward, b.reward), (a.done, b.done), (a.player, b.player), (a.env_info(), b.env_info())):
            assert torch.equal(x, y), auto
        a.close(); b.close()


def test_multi_step_launch_falls_back_where_it_does_not_apply():
    """A call of ONE step: one launch; rings of nine (pointers in a device table) and of eight sets (pointers in the kernel arguments): multi-step
    launches with the results of one launch per step; `chains='auto'` prefers the multi-step launch to two chains where it applies."""
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv('micro', 300, seed=8, auto_reset=True)
    b = VecStrategoEnv('micro', 300, seed=8, auto_reset=True)
    b.set_multi_step(False)
    a.reset(); b.reset()
    a.rollout_steps(1); b.rollout_steps(1)
    assert a.last_launch_kind == _lib.LAUNCH_WAVE
    a.rollout_steps(7, chains='auto'); b.rollout_steps(7)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP
    assert torch.equal(a.obs, b.obs) and torch.equal(a.mask, b.mask) and torch.equal(a.env_info(), b.env_info())
    a.alloc_output_ring(9); b.alloc_output_ring(9)
    a.rollout_steps(20, ring=True); b.rollout_steps(20, ring=True)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP               # nine sets: their pointers travel in a device table
    for (oa, ma, _), (ob, mb, _) in zip(a._ring, b._ring):
        assert torch.equal(oa, ob) and torch.equal(ma, mb)
    a.alloc_output_ring(8); b.alloc_output_ring(8)
    a.rollout_steps(20, ring=True); b.rollout_steps(20, ring=True)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP               # eight fit
    for (oa, ma, _), (ob, mb, _) in zip(a._ring, b._ring):
        assert torch.equal(oa, ob) and torch.equal(ma, mb)
    assert torch.equal(a.next_actions, b.next_actions) and torch.equal(a.env_info(), b.env_info())
    a.close(); b.close()
