"""The multi-step launch of the wave-per-game kernels (stratego_env_amd/csrc/sgx_step.h: steps_kernel): all steps of an sgx_step_n /
sgx_step_ring call in ONE launch -- a workgroup stages its games once, every wave plays its game step after step with the boards in LDS
and the scalars in registers, the record goes back to HBM once.  It must leave exactly what one launch per step leaves: every output of
the last step, every output set of a ring, rewards / flags, the next draw, the int64 states and counters -- on every board size and
observation kind it covers, from garbage first actions, across the 256-step chunking, with auto-reset and without."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [('barrage', 700, 70), ('standard', 300, 130), ('octa_barrage', 513, 64), ('medium', 1000, 90), ('fives', 777, 61), ('standard2', 40, 50),
         ('short_barrage', 2048, 120)]


def _pair(name, n, **kw):
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=4711, env_id_offset=3, auto_reset=kw.pop('auto_reset', True), **kw)
    b = VecStrategoEnv(name, n, seed=4711, env_id_offset=3, auto_reset=a.auto_reset, **kw)
    b.set_multi_step(False)
    a.reset(); b.reset()
    return a, b


def _same(a, b, what):
    import torch
    for x, y, w in ((a.obs, b.obs, 'obs'), (a.mask, b.mask, 'mask'), (a.reward, b.reward, 'reward'), (a.done, b.done, 'done'), (a.player, b.player, 'player'),
                    (a.next_actions, b.next_actions, 'next action'), (a.ending_invalid, b.ending_invalid, 'ending_invalid'), (a.env_info(), b.env_info(), 'info')):
        assert torch.equal(x, y), (what, w)
    if a.fobs is not None:
        assert torch.equal(a.fobs, b.fobs), (what, 'fobs')
    sa, pa = a.export_state()
    sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb), (what, 'state')


@pytest.mark.parametrize('name,n,T', CASES)
def test_multi_step_launch_equals_one_launch_per_step(name, n, T):
    import torch
    from stratego_env_amd import _lib
    a, b = _pair(name, n)
    a.sample_valid_actions(); b.sample_valid_actions()
    bad = torch.arange(n, device=a.device) % 5 == 2                  # some envs start from a garbage action
    for e in (a, b):
        e.next_actions[bad] = -9
        e._next_actions_fresh = True
    a.rollout_steps(T)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE          # (the test must not pass on another kernel)
    for _ in range(T):
        b.rollout_steps(1)
    assert b.last_launch_kind == _lib.LAUNCH_WAVE
    _same(a, b, (name, 'in place'))
    a.alloc_output_ring(3); b.alloc_output_ring(3)
    a.rollout_steps(T + 1, ring=True); b.rollout_steps(T + 1, ring=True)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
    for (oa, ma, _), (ob, mb, _) in zip(a._ring, b._ring):
        assert torch.equal(oa, ob) and torch.equal(ma, mb), (name, 'ring')
    _same(a, b, (name, 'ring'))
    # rollouts without observation / without any output (the no-observation kind), then with everything again
    a.rollout_steps(9, emit_obs=False); b.rollout_steps(9, emit_obs=False)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE and torch.equal(a.mask, b.mask)
    a.rollout_steps(11, emit_obs=False, emit_mask=False); b.rollout_steps(11, emit_obs=False, emit_mask=False)
    a.rollout_steps(4); b.rollout_steps(4)
    _same(a, b, (name, 'after logic-only rollouts'))
    assert int(a.invalid_action.sum()) == int(b.invalid_action.sum())
    a.close(); b.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('name,n', [('barrage', 1024), ('standard', 512), ('octa_barrage', 768), ('medium', 1024), ('fives', 1024), ('standard2', 64),
                                    ('barrage', 1021)])
def test_barrier_per_step_leaves_the_same_results(name, n):
    """sgx_set_steps_barrier: the waves of a workgroup in step (1 = in every multi-step launch; -1 = the default: with more than 8 output sets and
    float32 observations) against waves that drift (0) -- in place, into a ring of 12 sets, into a 20-slot trajectory buffer, with both
    observations, with compact outputs and without any output: the same bytes.  (n = 1021: the last workgroup is partly empty -- the library
    leaves the barrier out of that launch, whatever the mode.)"""
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.vec_env import VecStrategoEnv
    for kw in ({}, {'full_obs': True}, {'compact_outputs': True}):
        envs = []
        for mode in (0, 1, -1):
            e = VecStrategoEnv(name, n, seed=99, env_id_offset=7, auto_reset=True, **kw)
            e.set_steps_barrier(mode)
            e.reset(); e.sample_valid_actions()
            envs.append(e)
        base = envs[0]
        compact = bool(kw.get('compact_outputs'))

        def same(a, b, what):
            if not compact:
                return _same(a, b, what)
            # (compact records hold stale bytes behind their last entry: what they DECODE to is the contract)
            assert torch.equal(a.decode_mask(), b.decode_mask()) and torch.equal(a.decode_obs(), b.decode_obs()), what
            for x, y in ((a.reward, b.reward), (a.done, b.done), (a.player, b.player), (a.next_actions, b.next_actions), (a.env_info(), b.env_info())):
                assert torch.equal(x, y), what
        for e in envs:
            e.rollout_steps(33)
            assert e.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
        for e in envs[1:]:
            same(base, e, (name, kw, 'in place'))
        for e in envs:
            e.alloc_output_ring(12)
            e.rollout_steps(29, ring=True)
            assert e.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
        for e in envs[1:]:
            for (oa, ma, fa), (ob, mb, fb) in zip(base._ring, e._ring):
                assert compact or (torch.equal(oa, ob) and torch.equal(ma, mb) and (fa is None or torch.equal(fa, fb))), (name, kw, 'ring of 12')
            same(base, e, (name, kw, 'ring of 12'))
        trajs = []
        for e in envs:
            t = e.alloc_trajectory(20)
            e.rollout_trajectory(27, t)
            trajs.append(t)
        for e, t in zip(envs[1:], trajs[1:]):
            for k in trajs[0]:
                if not (compact and k in ('obs', 'mask')):
                    assert torch.equal(trajs[0][k].view(torch.uint8), t[k].view(torch.uint8)), (name, kw, 'trajectory', k)
            same(base, e, (name, kw, 'trajectory'))
        for e in envs:
            e.rollout_steps(13, emit_obs=False, emit_mask=False)
            e.rollout_steps(3)
        for e in envs[1:]:
            same(base, e, (name, kw, 'after a logic-only rollout'))
        for e in envs:
            e.close()
    with pytest.raises(Exception):
        bad = VecStrategoEnv(name, 8, seed=1)
        try:
            bad.set_steps_barrier(2)
        finally:
            bad.close()


def test_multi_step_launch_in_chunks_and_against_the_oracle():
    """600 steps = three launches (256 + 256 + 88); sampled games against the oracle's digests of the last step and its counters."""
    import torch
    from oracle import oracle as orc
    from stratego_env_amd import _lib, setups as S
    from stratego_env_amd.vec_env import VecStrategoEnv
    from tests.helpers import oracle_cvariant
    N, T, seed = 4096 + 5, 600, 0xABCD
    env = VecStrategoEnv('barrage', N, seed=seed, auto_reset=True)
    env.reset()
    env.rollout_steps(T)
    assert env.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE and int(env.invalid_action.sum()) == 0
    cv = oracle_cvariant('barrage', setups=S.load_setup_table('barrage'))
    ids = [0, 1, 7, 8, 4095, 4096, N - 1]
    idx = torch.tensor(ids, device=env.device)
    mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
    rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
    ei, info = env.ending_invalid[idx].cpu().numpy(), env.env_info()[idx].cpu().numpy()
    for i, e in enumerate(ids):
        r = orc.rollout_ex(cv, seed, e, 1, T, threads=1)
        assert int(r['last_digests'][0]) == orc.step_digest(mk[i], ob[i], rw[i], dn[i], pl[i], ei[i]), e
        assert np.array_equal(r['info'][0], info[i]), e
    env.close()


@pytest.mark.parametrize('kw,what', [({'full_obs': True}, 'BOTH_OBSERVATIONS'), ({'compact_outputs': True}, 'compact outputs'),
                                     ({'final_obs': True}, 'terminal observations'), ({'auto_reset': False}, 'no auto-reset')])
def test_multi_step_launch_other_kinds(kw, what):
    import torch
    from stratego_env_amd import _lib
    a, b = _pair('short_barrage', 1500, **dict(kw))
    a.rollout_steps(150); b.rollout_steps(150)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE and b.last_launch_kind == _lib.LAUNCH_WAVE, what
    if a.compact:
        # (the compact records themselves hold stale bytes behind their last entry: what they DECODE to is the contract)
        assert torch.equal(a.mask, b.mask) and torch.equal(a.decode_mask(), b.decode_mask()) and torch.equal(a.decode_obs(), b.decode_obs())
        for x, y in ((a.reward, b.reward), (a.done, b.done), (a.next_actions, b.next_actions), (a.env_info(), b.env_info())):
            assert torch.equal(x, y), what
    else:
        _same(a, b, what)
    if a.final_obs is not None:
        assert torch.equal(a.final_obs, b.final_obs) and bool(a.final_obs.abs().sum() > 0)
    a.close(); b.close()


def test_multi_step_launch_is_not_taken_where_it_does_not_apply():
    import torch
    from stratego_env_amd import _lib
    a, b = _pair('barrage', 256, obs_channel_mode='original')
    a.rollout_steps(20); b.rollout_steps(20)
    assert a.last_launch_kind == _lib.LAUNCH_WAVE                     # 'original' channels: one launch per step
    assert torch.equal(a.obs, b.obs) and torch.equal(a.env_info(), b.env_info())
    a.close(); b.close()
    a, b = _pair('barrage', 256)
    a.alloc_output_ring(9); b.alloc_output_ring(9)
    a.rollout_steps(20, ring=True); b.rollout_steps(20, ring=True)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE          # nine output sets: more than the kernel arguments hold -- a device table of pointers
    for (oa, ma, _), (ob, mb, _) in zip(a._ring, b._ring):
        assert torch.equal(oa, ob) and torch.equal(ma, mb)
    a.rollout_steps(1)
    assert a.last_launch_kind == _lib.LAUNCH_WAVE                     # a single step
    a.rollout_steps(6, chains=2)
    assert a.last_launch_kind == _lib.LAUNCH_WAVE                     # explicit chains of launches
    a.close(); b.close()


@pytest.mark.parametrize('name,n', [('fives', 777), ('medium', 1000), ('octa_barrage', 513), ('barrage', 300)])
def test_auto_chains_take_the_multi_step_launch_on_every_wave_board(name, n):
    """sgx_rollout(chains = 0): the multi-step launch wherever the call is eligible -- also on boards of 17 .. 36 cells, where two chains of
    per-step launches were the rule until the kernel's parameter reads became scalar loads -- with the results of one launch per step."""
    from stratego_env_amd import _lib
    a, b = _pair(name, n)
    a.rollout_steps(23, chains='auto'); b.rollout_steps(23)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE and b.last_launch_kind == _lib.LAUNCH_WAVE
    _same(a, b, name)
    a.close(); b.close()


@pytest.mark.parametrize('name,n,T', [('barrage', 65536, 48), ('standard', 262144, 24), ('micro', 65536, 64), ('fives', 65536 + 17, 40), ('standard2', 32768 + 5, 16)])
def test_multi_step_launch_at_the_baseline_sizes_equals_one_launch_per_step(name, n, T):
    """BASELINE configs 2 / 3 / 4 (and two ragged odd boards) at full size: ONE rollout call (multi-step launches: steps_kernel, on Micro
    lane_steps_kernel) against the same steps as one launch per step -- every output tensor of the last step, rewards / flags, the next draw
    and the int64 state of EVERY game.  (The per-step kernel itself is pinned to the oracle at these sizes by tests/test_gpu_parity.py.)"""
    import torch
    from stratego_env_amd import _lib
    a, b = _pair(name, n)
    a.rollout_steps(T); b.rollout_steps(T)
    assert a.last_launch_kind in (_lib.LAUNCH_MULTI_STEP_WAVE, _lib.LAUNCH_MULTI_STEP) and b.last_launch_kind in (_lib.LAUNCH_WAVE, _lib.LAUNCH_LANE)
    _same(a, b, name)
    assert int(a.invalid_action.sum()) == 0 and int(a.env_info()[:, 0].min()) >= 0
    a.close(); b.close()
    torch.cuda.empty_cache()


def test_logic_only_rollouts_on_micro_stay_on_the_lane_kernel():
    """A rollout WITHOUT an observation on the 3x4 board: the lane-per-game kernel (one launch per step) stays ahead of the wave-per-game
    kernel's multi-step launch there (tools/noobs_small_ab.py: 13-15 against 16-18 us per step of 65,536 games), so that launch leaves
    these calls to it; with the lane kernel switched off the same call is a multi-step launch -- same results either way.  On 4x4 the
    multi-step launch is the faster one and is taken."""
    import torch
    from stratego_env_amd import _lib
    a, b = _pair('micro', 3000)
    b.set_multi_step(True)
    b.set_lane_kernel(False)
    for emit_mask in (True, False):
        a.rollout_steps(9, emit_obs=False, emit_mask=emit_mask); b.rollout_steps(9, emit_obs=False, emit_mask=emit_mask)
        assert a.last_launch_kind == _lib.LAUNCH_LANE and b.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
        assert torch.equal(a.env_info(), b.env_info()) and torch.equal(a.next_actions, b.next_actions) and torch.equal(a.reward, b.reward)
        if emit_mask:
            assert torch.equal(a.mask, b.mask)
    a.rollout_steps(5); b.rollout_steps(5)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP and b.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE
    _same(a, b, 'micro')
    a.close(); b.close()
    a, b = _pair('tiny', 2000)
    a.rollout_steps(9, emit_obs=False); b.rollout_steps(9, emit_obs=False)
    assert a.last_launch_kind == _lib.LAUNCH_MULTI_STEP_WAVE and b.last_launch_kind == _lib.LAUNCH_LANE
    assert torch.equal(a.mask, b.mask) and torch.equal(a.env_info(), b.env_info()) and torch.equal(a.next_actions, b.next_actions)
    a.close(); b.close()
