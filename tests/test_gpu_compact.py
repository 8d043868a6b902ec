"""Opt-in compact outputs (SGX_STEP_COMPACT_OBS / _MASK): the step writes its 4-bit code buffer and the mask bits instead of the float32
observation and the mask bytes; sgx_decode_obs / sgx_decode_mask expand them.  The decoded tensors must be byte-identical to what a
non-compact env writes for the same games -- on every board size (aligned and odd), with uncoded entries (Standard mid-game, piece
sets that normalise to thirds), through step / observe / step_n / rollout chains / the ring -- and the game trajectories must not
depend on the output format."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,n,steps', [('barrage', 300, 60), ('standard', 64, 260), ('fives', 257, 50), ('standard2', 20, 120),
                                          ('medium', 200, 80), ('octa_barrage', 100, 60), ('tiny', 500, 60), ('micro', 1000, 40)])
def test_decoded_compact_outputs_are_byte_identical(name, n, steps):
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    a = VecStrategoEnv(name, n, seed=606, env_id_offset=5, auto_reset=True)
    b = VecStrategoEnv(name, n, seed=606, env_id_offset=5, auto_reset=True, compact_outputs=True)
    v = a.variant
    assert b.obs.shape == (n, b.compact_obs_stride) and b.obs.dtype == torch.uint8 and b.compact_obs_stride % 128 == 0
    assert b.compact_obs_stride >= (v.cells * 67 + 1) // 2 + 16 and b.compact_obs_stride < v.cells * 67 * 4 / 5
    assert b.mask.shape == (n, b.compact_mask_words) and b.compact_mask_words * 32 >= v.num_spatial_actions
    a.reset(); b.reset()
    uncoded = 0
    for t in range(steps):
        assert torch.equal(b.decode_obs(), a.obs), (name, t, 'obs')
        assert torch.equal(b.decode_mask(), a.mask), (name, t, 'mask')
        nib_bytes = ((v.cells * 67 + 1) // 2 + 15) & ~15
        uncoded += int(b.obs[:, nib_bytes:nib_bytes + 4].contiguous().view(torch.int32).sum())
        # the bit mask itself: bit i of the words = mask byte i
        if t % 16 == 0:
            words = b.mask.cpu().numpy().astype(np.uint32)
            bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)[:, :v.num_spatial_actions]
            assert np.array_equal(bits.astype(np.uint8), a.mask.reshape(n, -1).cpu().numpy())
        a.rollout_step(); b.rollout_step()
        for x, y in ((a.reward, b.reward), (a.done, b.done), (a.player, b.player), (a.next_actions, b.next_actions), (a.invalid_action, b.invalid_action)):
            assert torch.equal(x, y), (name, t)
    sa, pa = a.export_state()
    sb, pb = b.export_state()
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    if name == 'standard':
        assert uncoded > 0                       # captured miners / majors / bombs: entries without a code went through the side list
    # fused multi-step paths write the same compact records
    a.rollout_steps(7); b.rollout_steps(7)
    assert torch.equal(b.decode_obs(), a.obs) and torch.equal(b.decode_mask(), a.mask)
    a.rollout_steps(6, chains=2); b.rollout_steps(6, chains=2)
    assert torch.equal(b.decode_obs(), a.obs) and torch.equal(b.decode_mask(), a.mask)
    a.alloc_output_ring(3); b.alloc_output_ring(3)
    a.rollout_steps(5, ring=True); b.rollout_steps(5, ring=True)
    assert torch.equal(b.decode_obs(), a.obs) and torch.equal(b.decode_mask(), a.mask)
    a.close(); b.close()


def test_compact_outputs_with_thirds_and_refusals():
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.config import custom_variant
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = custom_variant(6, 6, max_turns=90, piece_counts=(0, 0, 0, 3, 3, 0, 0, 0, 0, 0, 1, 0), initial_state_usable_rows=2, name='thirds66')
    a = VecStrategoEnv(v, 128, seed=9, auto_reset=True)
    b = VecStrategoEnv(v, 128, seed=9, auto_reset=True, compact_outputs=True)
    a.reset(); b.reset()
    seen = 0
    for t in range(120):
        a.rollout_step(); b.rollout_step()
        assert torch.equal(b.decode_obs(), a.obs), t
        nib_bytes = ((36 * 67 + 1) // 2 + 15) & ~15
        seen += int(b.obs[:, nib_bytes:nib_bytes + 4].contiguous().view(torch.int32).sum())
    assert seen > 0
    a.close(); b.close()
    with pytest.raises(ValueError):
        VecStrategoEnv('barrage', 8, compact_outputs=True, full_obs=True)
    with pytest.raises(ValueError):
        VecStrategoEnv('barrage', 8, compact_outputs=True, obs_channel_mode='original')
    env = VecStrategoEnv('barrage', 8, seed=1, compact_outputs=True)
    env.reset()
    with pytest.raises(_lib.SgxError):                         # state-coordinate masks are not compact
        env.step(env.sample_valid_actions(), flags=_lib.STEP_MASK_1D)
    with pytest.raises(ValueError):
        env.tune_placement()
    env.close()
