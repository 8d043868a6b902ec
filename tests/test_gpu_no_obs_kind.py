"""Launches without any observation pointer run an instantiation of their own (KIND 8: no code buffer, no observation tables in LDS,
8 waves per SIMD).  Its game logic, mask and sampler are the same source as the observing kernels'; this file pins that the RESULTS are
too: a twin env that emits observations (the kernels the parity suites compare with the oracle) must see the same masks, rewards,
flags, sampled actions and int64 states step by step."""
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = [('barrage', 512), ('standard', 256), ('octa_barrage', 512), ('medium', 512), ('fives', 777), ('tiny', 1000), ('micro', 1000),
            ('standard2', 96)]


@pytest.mark.parametrize('name,n', VARIANTS)
@pytest.mark.parametrize('emit_mask', [True, False])
def test_steps_without_observation_equal_steps_with_one(name, n, emit_mask):
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=11, auto_reset=True)
    b = VecStrategoEnv(name, n, seed=11, auto_reset=True)
    b.set_lane_kernel(0)                     # toy boards: the wave-per-game no-observation kind, not the lane kernel (tested on its own)
    a.reset()
    b.reset()
    a.sample_valid_actions()
    b.sample_valid_actions()
    for t in range(70):
        assert torch.equal(a.next_actions, b.next_actions), (name, t)
        a.step(a.next_actions, want_next_actions=True)
        b.step(b.next_actions, want_next_actions=True, emit_obs=False, emit_mask=emit_mask)
        for x, y, what in ((a.reward, b.reward, 'reward'), (a.done, b.done, 'done'), (a.player, b.player, 'player'),
                           (a.invalid_action, b.invalid_action, 'invalid')):
            assert torch.equal(x, y), (name, t, what)
        if emit_mask:
            assert torch.equal(a.mask, b.mask), (name, t)
    sa, pa = a.export_state()
    sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    # the state-preserving observe launch without an observation: the same mask
    b.observe(emit_obs=False)
    a.observe()
    assert torch.equal(a.mask, b.mask)
    assert int(a.done.sum()) >= 0 and int(a.invalid_action.sum()) == 0
    a.close()
    b.close()


def test_garbage_actions_without_observation():
    """Injected garbage (out-of-range indices, the no-op channel, other cells' moves): accept / reject decisions and what a rejected
    action leaves behind are the same with and without an observation."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    n = 2048
    a = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    b = VecStrategoEnv('barrage', n, seed=5, auto_reset=True)
    a.reset()
    b.reset()
    g = torch.Generator(device='cpu').manual_seed(3)
    for t in range(40):
        a.sample_valid_actions()
        acts = a.next_actions.clone()
        junk = torch.randint(-50, 3700 + 50, (n,), generator=g, dtype=torch.int32).to(acts.device)
        use = (torch.rand(n, generator=g) < 0.3).to(acts.device)
        acts = torch.where(use, junk, acts)
        a.step(acts)
        b.step(acts, emit_obs=False)
        assert torch.equal(a.invalid_action, b.invalid_action) and torch.equal(a.mask, b.mask) and torch.equal(a.reward, b.reward)
        assert torch.equal(a.done, b.done) and torch.equal(a.player, b.player)
    assert int(a.invalid_action.sum()) > 0
    sa, _ = a.export_state()
    sb, _ = b.export_state()
    assert torch.equal(sa, sb)
    a.close()
    b.close()


@pytest.mark.parametrize('name,n', VARIANTS + [('short_barrage', 2048)])
def test_steps_that_only_ask_whether_a_move_exists(name, n):
    """A launch that wants neither observation nor mask nor a next action (search expansion, logic-only steps) does not build the mask:
    it only asks WHETHER the next mover has a move (gen_mask<..., ANY>: one cell per ray, further only behind a vetoed cell).  Endings by
    a stuck opponent, rewards, flags and states must be those of the full step -- over whole games on every board size, with garbage
    actions and the no-op (whose legality is the same question) injected."""
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=23, auto_reset=True)
    b = VecStrategoEnv(name, n, seed=23, auto_reset=True)
    b.set_lane_kernel(0)
    a.reset()
    b.reset()
    na = a.variant.num_spatial_actions
    g = torch.Generator(device='cpu').manual_seed(9)
    ended = 0
    for t in range(120):
        a.sample_valid_actions()
        acts = a.next_actions.clone()
        junk = torch.randint(-3, na + 3, (n,), generator=g, dtype=torch.int32).to(acts.device)
        junk = torch.where((torch.rand(n, generator=g) < 0.5).to(acts.device), torch.full_like(junk, a.K - 1), junk)     # the spatial no-op [0,0,K-1]
        acts = torch.where((torch.rand(n, generator=g) < 0.1).to(acts.device), junk, acts)
        a.step(acts)
        b.step(acts, emit_obs=False, emit_mask=False, want_next_actions=False)
        for x, y, what in ((a.reward, b.reward, 'reward'), (a.done, b.done, 'done'), (a.player, b.player, 'player'),
                           (a.invalid_action, b.invalid_action, 'invalid'), (a.ending_invalid, b.ending_invalid, 'ending_invalid')):
            assert torch.equal(x, y), (name, t, what)
        ended += int(a.done.sum())
        if t % 30 == 29:
            assert torch.equal(a.env_info(), b.env_info()), (name, t)
    assert ended > 0
    sa, pa = a.export_state()
    sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    a.close()
    b.close()


@pytest.mark.parametrize('name,n', [('barrage', 1003), ('standard', 517), ('octa_barrage', 1000), ('medium', 999), ('short_barrage', 2050)])
def test_two_games_per_wave_equal_one_game_per_wave(name, n, monkeypatch):
    """Launches without an observation play TWO games per wave on boards of 33 .. 128 cells (Geo<R, C, 2>, the default); sgx_set_half_wave(h, 0)
    (or SGX_HALF_WAVE=0 at handle creation: the pools of the functional API below) gives one game per wave.  Ragged batch sizes (partial workgroups, a wave with one game), per-step and multi-step
    launches, mask-only and logic-only, search expansion pool to pool: identical masks, rewards, flags, draws and int64 states.  (Both are
    compared with the oracle through the suites of the no-observation kind; this test pins the two layouts to each other on every board.)"""
    import torch
    from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=21, auto_reset=True)
    a.set_half_wave(False)                                              # one game per wave (sgx_set_half_wave)
    b = VecStrategoEnv(name, n, seed=21, auto_reset=True)               # the default: two
    a.reset(); b.reset()
    a.sample_valid_actions(); b.sample_valid_actions()
    bad = torch.arange(n, device=a.device) % 9 == 4
    for e in (a, b):
        e.next_actions[bad] = -3
        e._next_actions_fresh = True
    for multi in (False, True):
        a.set_multi_step(multi); b.set_multi_step(multi)
        for kw in ({'emit_obs': False}, {'emit_obs': False, 'emit_mask': False}):
            a.rollout_steps(37, **kw); b.rollout_steps(37, **kw)
            for x, y, w in ((a.reward, b.reward, 'reward'), (a.done, b.done, 'done'), (a.player, b.player, 'player'), (a.next_actions, b.next_actions, 'draw'),
                            (a.invalid_action, b.invalid_action, 'invalid'), (a.ending_invalid, b.ending_invalid, 'ending'), (a.env_info(), b.env_info(), 'info')):
                assert torch.equal(x, y), (name, multi, kw, w)
            if kw.get('emit_mask', True):
                assert torch.equal(a.mask, b.mask), (name, multi)
    sa, pa = a.export_state(); sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    a.observe(emit_obs=False); b.observe(emit_obs=False)
    assert torch.equal(a.mask, b.mask)
    # search expansion (sgx_expand: the MAPPED no-observation instantiation) from these positions, both layouts
    kids = []
    for mode in ('0', '1'):
        monkeypatch.setenv('SGX_HALF_WAVE', mode)
        pe = BatchedStrategoProceduralEnv(name, n)
        m1 = pe.get_valid_moves_as_1d_mask(sa, pa)
        acts = torch.argmax((m1 != 0).to(torch.int8), dim=1).to(torch.int32)
        acts[::7] = 3                                                   # some invalid ones: the child is a copy of the parent
        nodes, pool = pe.pack(sa, pa), pe.new_packed()
        valid, movers = pool.expand(nodes, acts)
        kids.append((valid.clone(), movers.clone()) + tuple(pool.unpack()) + (m1,))
    for x, y in zip(kids[0], kids[1]):
        assert torch.equal(x, y), name
    a.close(); b.close()
