"""sgx_step_sync on a handful of games: the single_kernel path (one workgroup per game, mask and observations emitted by all eight
waves, completion polled in host-mapped memory) against sgx_step on a twin handle -- every output of every step, byte for byte
(sgx_step itself is compared with the oracle in test_gpu_parity.py; the N = 1 facade, which steps through sgx_step_sync, replays
the reference's recorded episodes in test_gpu_facade.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _outputs(env):
    names = ['obs', 'mask', 'reward', 'done', 'player', 'invalid_action', 'ending_invalid', 'final_obs']
    if env.fobs is not None:
        names += ['fobs', 'final_fobs']
    return {n: getattr(env, n).cpu().numpy().copy() for n in names}


@pytest.mark.parametrize('name,n_envs,n_steps,both', [
    ('barrage', 1, 500, False), ('barrage', 8, 400, True), ('barrage', 5, 300, False), ('standard', 3, 900, False), ('standard', 2, 700, True),
    ('octa_barrage', 8, 400, True), ('medium', 7, 300, False), ('short_barrage', 1, 300, True),
    ('fives', 4, 150, False), ('micro', 8, 100, False), ('barrage', 9, 100, False),     # not eligible: the ordinary kernel + a stream wait
])
def test_step_sync_equals_step(name, n_envs, n_steps, both):
    import torch
    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[name]
    kw = dict(seed=0x51A6E + n_envs, env_id_offset=77, auto_reset=True, final_obs=True, full_obs=both)
    a_env, b_env = VecStrategoEnv(name, n_envs, **kw), VecStrategoEnv(name, n_envs, **kw)
    a_env.reset()
    b_env.reset()
    rs = np.random.RandomState(n_envs + len(name))
    NA = v.num_spatial_actions
    acts = a_env.sample_valid_actions().cpu().numpy()
    ended = 0
    for t in range(n_steps):
        for e in range(n_envs):
            if rs.rand() < 0.08:
                acts[e] = int(rs.choice([rs.randint(NA), -1, NA, NA + 5, rs.randint(v.cells) * v.spatial_channels + v.spatial_channels - 1]))
        at = torch.from_numpy(acts.astype(np.int32))
        a_env.step(at, want_next_actions=True)
        b_env.step_sync(at)                               # (no torch.cuda.synchronize: the call itself waits)
        b_out = {k: val.cpu().numpy() for k, val in (('obs', b_env.obs), ('mask', b_env.mask))}
        a_out = _outputs(a_env)
        b_all = _outputs(b_env)
        for k in ('reward', 'done', 'player', 'invalid_action', 'ending_invalid', 'mask', 'obs') + (('fobs',) if both else ()):
            assert a_out[k].tobytes() == b_all[k].tobytes(), (name, n_envs, t, k)
        assert b_out['obs'].tobytes() == a_out['obs'].tobytes()
        done = a_out['done'].astype(bool)
        if done.any():                                      # terminal observations are defined for the games that ended in this step
            ended += int(done.sum())
            assert a_out['final_obs'][done].tobytes() == b_all['final_obs'][done].tobytes(), (name, t, 'final_obs')
            if both:
                assert a_out['final_fobs'][done].tobytes() == b_all['final_fobs'][done].tobytes(), (name, t, 'final_fobs')
        acts = a_env.next_actions.cpu().numpy().copy()
    sa, pa = a_env.export_state()
    sb, pb = b_env.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    assert ended > 0 or name not in ('short_barrage', 'micro', 'octa_barrage'), 'no game ended: terminal outputs were not compared'
    a_env.close()
    b_env.close()


def test_step_sync_outputs_in_host_memory_are_complete_on_return():
    """The facade's arrangement: the action word and every output live in device-addressable pinned host memory (sgx_host_alloc)
    and are read by the host right after the call, with no synchronisation of its own -- 2,000 steps against a twin that steps through
    sgx_step into device tensors."""
    import ctypes as C
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS['barrage']
    kw = dict(seed=99, auto_reset=True)
    a_env, b_env = VecStrategoEnv('barrage', 1, **kw), VecStrategoEnv('barrage', 1, **kw)
    a_env.reset()
    b_env.reset()
    L = b_env._L
    obs_b, mask_b = v.cells * 67 * 4, v.num_spatial_actions
    total = obs_b + mask_b + 64 + 256
    hp, dp = C.c_void_p(), C.c_void_p()
    _lib.check(L.sgx_host_alloc(b_env._h, total, C.byref(hp), C.byref(dp)), L)
    raw = np.ctypeslib.as_array((C.c_uint8 * total).from_address(hp.value))
    off_mask, off_misc = obs_b, (obs_b + mask_b + 63) & ~63
    io = _lib.SgxStepIO()
    io.actions_dev = dp.value + off_misc + 32
    io.obs_dev, io.mask_dev = dp.value, dp.value + off_mask
    io.reward_dev, io.done_dev, io.player_dev = dp.value + off_misc, dp.value + off_misc + 8, dp.value + off_misc + 9
    io.invalid_action_dev, io.ending_invalid_dev = dp.value + off_misc + 10, dp.value + off_misc + 11
    io.auto_reset = 1
    io.flags = b_env._mode_flags
    act_word = raw[off_misc + 32:off_misc + 36].view(np.int32)
    acts = a_env.sample_valid_actions()
    for t in range(2000):
        act_word[0] = int(acts[0])
        a_env.step(acts, want_next_actions=True)
        _lib.check(L.sgx_step_sync(b_env._h, C.byref(io), b_env._stream()), L)
        got_obs, got_mask = raw[:obs_b].copy(), raw[off_mask:off_mask + mask_b].copy()      # read at once: no synchronisation here
        got_misc = raw[off_misc:off_misc + 12].copy()
        assert got_obs.tobytes() == a_env.obs.cpu().numpy().tobytes(), t
        assert got_mask.tobytes() == a_env.mask.cpu().numpy().tobytes(), t
        assert got_misc[:8].tobytes() == a_env.reward.cpu().numpy().tobytes()
        assert (got_misc[8], got_misc[9].view(np.int8)) == (int(a_env.done[0]), int(a_env.player[0]))
        acts = a_env.next_actions.clone()
    torch.cuda.synchronize()
    _lib.check(L.sgx_host_free(b_env._h, hp), L)
    a_env.close()
    b_env.close()
