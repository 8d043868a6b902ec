"""VecStrategoEnv: N concurrent games on one MI355X, torch tensors in / out.

The batched counterpart of StrategoMultiAgentEnv (reference stratego_multiagent_env.py:316-834): every
call goes through the C ABI in include/stratego_mi355x.h to the HIP kernels; PyTorch only owns the device
buffers and the stream.  Outputs are written in place into preallocated tensors:

    obs      float32 [N, R, C, 67]   normalised partial observation of each env's next mover
    fobs     float32 [N, R, C, 79]   normalised fully-observable observation (only with full_obs=True)
                                     (32 / 33 channels with obs_channel_mode='original')
    mask     uint8   [N, R, C, K]    valid-actions mask of the next mover (flat index = action)
    reward   float32 [N, 2]          rewards of player +1 / -1 (non-zero only when done)
    done     uint8   [N]
    player   int8    [N]             next mover (+1 / -1)
    invalid_action uint8 [N]         the reference would have raised ValueError; env unchanged
    ending_invalid uint8 [N]         game ended by max_turns ('game_result_was_invalid')

There is no CPU fallback: construction fails if the HIP library is missing or no GPU is visible.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

from . import _lib
from .config import (FO_OBS_CHANNELS, FO_OBS_CHANNELS_ORIGINAL, PO_OBS_CHANNELS, PO_OBS_CHANNELS_ORIGINAL, NUM_STATE_LAYERS,
                     get_variant)
from .setups import load_setup_table


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _OutputsOwner:
    """One sgx_alloc_outputs allocation.  It is freed when the LAST reference goes: the env holds one, and so does every tensor
    that views the buffers (torch keeps the object behind __cuda_array_interface__ alive as long as the tensor's storage), so a
    tensor handed out by step() / rollout_steps() / observe() stays valid after env.close(), `del env` or another
    tune_placement() -- like the torch-owned buffers it replaces."""

    def __init__(self, lib, out):
        self._L, self.out = lib, out

    def __del__(self):
        # Runs wherever the garbage collector drops the last tensor.  sgx_free_outputs selects the buffers' device for the free and
        # puts the calling thread's device back (DeviceGuard); at interpreter shutdown the HIP runtime may already be gone and the
        # process's memory is released anyway: nothing to do then.
        # (module globals may already be None at shutdown, and an import here would raise: everything inside the try)
        try:
            if sys is None or sys.is_finalizing():
                return
            self._L.sgx_free_outputs(None, C.byref(self.out))      # (works without the handle: the record names its device)
        except Exception:
            pass


class _DeviceBuffer:
    """A library-owned device allocation seen through __cuda_array_interface__ (zero-copy into a torch tensor)."""

    def __init__(self, ptr, shape, typestr, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2,
                                         'strides': None}


def _wrap_device(ptr, shape, dtype, device, owner):
    typestr = {torch.float32: '<f4', torch.uint8: '|u1'}[dtype]
    t = torch.as_tensor(_DeviceBuffer(ptr, shape, typestr, owner), device=device)
    assert t.data_ptr() == int(ptr) and t.dtype == dtype
    return t


class VecStrategoEnv:
    def __init__(self, version='barrage', num_envs=1, device=0, seed=0, env_id_offset=0, human_inits=None,
                 auto_reset=False, final_obs=False, full_obs=False, lib_path=None, obs_channel_mode='extended', compact_outputs=False,
                 outputs=True, placement=None):
        """human_inits: None = use the Gravon table when the variant has one (util.py:301-319), False = uniformly
        random back-row placement (util.py:33-53), True = require the table.
        obs_channel_mode: 'extended' (67 / 79 one-hot channels) or 'original' (the deprecated 32 / 33 value channels,
        maenv:67, 368-375).
        outputs=False: a pool of game records without observation / mask tensors (`obs`, `mask`, `fobs` are None): what
        snapshot() and the packed search pools of procedural_env hold -- 0.5 KB per Barrage game instead of 31 KB.
        placement: 'search' (default) = the first reset() of all envs moves obs / mask / fobs into library-owned buffers picked by the bounded
        placement search (tune_placement(): ~10 ms, at most min(8 GiB, a quarter of the free device memory) held beyond the buffers for a
        moment; which physical memory an allocation gets decides 0.76-0.93 against 0.95-1.04 of the roofline) -- only
        where there are placement classes (observations of more than 300 MB, not compact); 'plain' = torch.empty tensors, always.  None =
        the environment variable SGX_PLACEMENT, else 'search'.  (Use the tensors reset() / step() RETURN, or env.obs / env.mask read
        after the first reset(): that reset may replace the ones allocated here.)"""
        if obs_channel_mode not in ('extended', 'original'):
            raise ValueError("obs_channel_mode must be 'extended' or 'original'")
        placement = placement if placement is not None else (os.environ.get('SGX_PLACEMENT') or 'search')
        if placement not in ('plain', 'search'):
            raise ValueError("placement must be 'plain' or 'search' (SGX_PLACEMENT=%r)" % placement)
        self._placement_pending = placement == 'search'
        self.placement_report = None         # what the search of placement='search' saw (tune_placement's report)
        # compact_outputs=True (opt-in): `obs` is uint8 [N, compact_obs_stride] -- the 4-bit codes the float32 observation decodes from --
        # and `mask` int32 [N, compact_mask_words] -- one bit per action: 1/8 of the bytes per step.  decode_obs() / decode_mask() give the
        # contract tensors, byte-identical to a non-compact step (include/stratego_mi355x.h: SGX_STEP_COMPACT_OBS).
        self.compact = bool(compact_outputs)
        if self.compact and (final_obs or full_obs or obs_channel_mode != 'extended'):
            raise ValueError("compact_outputs comes with the 67-channel partial observation only (no final_obs / full_obs / original channels)")
        self.obs_channel_mode = obs_channel_mode
        self._mode_flags = _lib.STEP_ORIGINAL_CHANNELS if obs_channel_mode == 'original' else 0
        if compact_outputs:
            self._mode_flags |= _lib.STEP_COMPACT_OBS | _lib.STEP_COMPACT_MASK
        self.p_channels = PO_OBS_CHANNELS_ORIGINAL if obs_channel_mode == 'original' else PO_OBS_CHANNELS
        self.f_channels = FO_OBS_CHANNELS_ORIGINAL if obs_channel_mode == 'original' else FO_OBS_CHANNELS
        if not torch.cuda.is_available():
            raise _lib.SgxError("VecStrategoEnv needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
        self.variant = get_variant(version)
        v = self.variant
        self._L = _lib.load_for_geometry(v.rows, v.columns, lib_path)
        self.num_envs = int(num_envs)
        self.device = torch.device('cuda', device if isinstance(device, int) else torch.device(device).index or 0)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.env_id_offset = int(env_id_offset)
        self.auto_reset = bool(auto_reset)
        self.R, self.Cc, self.K = v.rows, v.columns, v.spatial_channels
        self._cfg = _lib.make_config(v)
        h = C.c_void_p()
        _lib.check(self._L.sgx_create(C.byref(self._cfg), self.num_envs, self.device.index, self.seed,
                                      self.env_id_offset, C.byref(h)), self._L)
        self._h = h
        if human_inits is None:
            human_inits = bool(v.human_inits)
        if human_inits:
            if not v.human_inits:
                raise ValueError("Human inits not supported with {} game version".format(v.name))  # util.py:310
            table = np.ascontiguousarray(load_setup_table(v.human_inits))
            _lib.check(self._L.sgx_set_setup_table(self._h, table.ctypes.data_as(C.c_void_p), table.shape[0]), self._L)
        self.human_inits = bool(human_inits)
        N, R, Cc, K, dev = self.num_envs, self.R, self.Cc, self.K, self.device
        self.has_outputs = bool(outputs)
        if not outputs:
            if compact_outputs or final_obs or full_obs:
                raise ValueError("outputs=False is a pool of records: no compact_outputs / final_obs / full_obs")
            self.obs = self.mask = None
        elif self.compact:
            self.compact_obs_stride = int(self._L.sgx_compact_obs_stride(self._h))
            self.compact_mask_words = int(self._L.sgx_compact_mask_words(self._h))
            self.obs = torch.empty((N, self.compact_obs_stride), dtype=torch.uint8, device=dev)
            self.mask = torch.empty((N, self.compact_mask_words), dtype=torch.int32, device=dev)
            self._mask_bytes = None
        else:
            self.obs = torch.empty((N, R, Cc, self.p_channels), dtype=torch.float32, device=dev)
            self.mask = torch.empty((N, R, Cc, K), dtype=torch.uint8, device=dev)
        self.reward = torch.zeros((N, 2), dtype=torch.float32, device=dev)
        self.done = torch.zeros((N,), dtype=torch.uint8, device=dev)
        self.player = torch.ones((N,), dtype=torch.int8, device=dev)
        self.invalid_action = torch.zeros((N,), dtype=torch.uint8, device=dev)
        self.ending_invalid = torch.zeros((N,), dtype=torch.uint8, device=dev)
        self.final_obs = torch.zeros((N, 2, R, Cc, self.p_channels), dtype=torch.float32, device=dev) if final_obs else None
        # fully-observable observation (ObservationModes FULLY_OBSERVABLE / BOTH_OBSERVATIONS), float32 [N,R,C,79]
        self.fobs = torch.empty((N, R, Cc, self.f_channels), dtype=torch.float32, device=dev) if full_obs else None
        self.final_fobs = (torch.zeros((N, 2, R, Cc, self.f_channels), dtype=torch.float32, device=dev)
                           if (full_obs and final_obs) else None)
        self.next_actions = torch.zeros((N,), dtype=torch.int32, device=dev)
        self._io = _lib.SgxStepIO()
        self._next_actions_fresh = False      # next_actions holds a draw for the CURRENT position of every env
        self._ring = self._ring_owners = self._ring_ios = None     # alloc_output_ring()
        self._ring_pos = 0
        self._outputs_owner = self._outputs = None                 # tune_placement()
        self.placement_peak_extra_bytes = 0

    # ---- lifecycle -----------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None):
            if getattr(self, '_outputs_owner', None) is not None:
                # library-owned buffers: the env lets go of its views too, so that the memory is freed now unless the caller still
                # holds a tensor of it (which then stays valid)
                self.obs = self.mask = self.fobs = None
            self._ring = self._ring_owners = None      # (tensors of the ring a caller still holds keep their buffers alive)
            self._release_outputs()
            self._L.sgx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def decode_obs(self, out=None):
        """compact_outputs only: the contract observation float32 [N,R,C,67] of the current compact `obs` (sgx_decode_obs)."""
        assert self.compact
        if out is None:
            out = torch.empty((self.num_envs, self.R, self.Cc, self.p_channels), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_decode_obs(self._h, _ptr(self.obs), _ptr(out), self._stream()), self._L)
        return out

    def decode_mask(self, out=None):
        """compact_outputs only: the contract mask uint8 [N,R,C,K] of the current bit mask (sgx_decode_mask)."""
        assert self.compact
        if out is None:
            out = torch.empty((self.num_envs, self.R, self.Cc, self.K), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_decode_mask(self._h, _ptr(self.mask), _ptr(out), self._stream()), self._L)
        return out

    def set_nt_stores(self, mode='auto'):
        """Store policy of the observation writes (sgx_set_nt_stores): 'auto' (by the launch's output size), False or True.
        Results are identical in every mode; it only moves the launch time."""
        if mode is None or mode == 'auto' or (mode == -1 and mode is not True):
            m = -1
        elif mode is True or mode is False or mode in (0, 1):
            m = int(mode)
        else:
            raise ValueError("set_nt_stores: 'auto', True or False")
        _lib.check(self._L.sgx_set_nt_stores(self._h, m), self._L)

    def set_lane_kernel(self, mode='auto'):
        """Kernel choice on boards of at most 16 cells (sgx_set_lane_kernel): 'auto' / True = one game per lane where the call is
        eligible, False = always the wave-per-game kernel.  Results are identical either way."""
        m = -1 if mode in ('auto', None) else int(bool(mode))
        _lib.check(self._L.sgx_set_lane_kernel(self._h, m), self._L)

    def set_steps_barrier(self, mode=-1):
        """Multi-step launches of the wave-per-game kernel: a barrier between the 8 waves of a workgroup before every step?  -1 (default) = where
        it pays (long rings / trajectory buffers of float32 observations: include/stratego_mi355x.h), 0 = never, 1 = always (sgx_set_steps_barrier).  Results
        are identical in every mode."""
        _lib.check(self._L.sgx_set_steps_barrier(self._h, int(mode)), self._L)

    def set_half_wave(self, on=True):
        """Launches without an observation (mask-only / logic-only steps and rollouts, search expansion) on boards of 33 .. 128 cells: True
        (default) = two games per wave, False = one (sgx_set_half_wave).  Results are identical either way."""
        _lib.check(self._L.sgx_set_half_wave(self._h, 1 if on else 0), self._L)

    def set_multi_step(self, on=True):
        """rollout_steps() on boards of at most 16 cells: True (default) = all steps of a call in one launch where eligible
        (sgx_set_multi_step: the games stay in registers, the logic of step t + 1 runs under the stores of step t); False = one launch
        per step.  Results are identical either way."""
        _lib.check(self._L.sgx_set_multi_step(self._h, 1 if on else 0), self._L)

    @property
    def last_launch_kind(self):
        """Which kernel the last step / observe launch was: _lib.LAUNCH_WAVE, LAUNCH_LANE, LAUNCH_MULTI_STEP or LAUNCH_MULTI_STEP_WAVE (sgx_last_launch_kind)."""
        return int(self._L.sgx_last_launch_kind(self._h))

    def set_xcd_skew(self, per_mille='auto'):
        """Shares of the eight XCDs in a launch (sgx_set_xcd_skew): 'auto' (100 per mille more for the even XCDs when the launch streams
        past the Infinity Cache, equal shares otherwise) or 0 .. 900.  Results are identical for every value."""
        _lib.check(self._L.sgx_set_xcd_skew(self._h, -1 if per_mille in ('auto', None) else int(per_mille)), self._L)

    def set_xcd_shares(self, per_mille=None):
        """Explicit shares of the eight XCDs (per mille of the mean share), None = back to the set_xcd_skew rule."""
        arr = None if per_mille is None else (C.c_int32 * 8)(*[int(x) for x in per_mille])
        _lib.check(self._L.sgx_set_xcd_shares(self._h, arr), self._L)

    # ---- API -----------------------------------------------------------------------------------------
    def reset(self, p1_maps=None, p2_maps=None, env_select=None):
        """Start new games (all envs, or those with env_select[i] != 0) and return (obs, mask, player).

        p1_maps / p2_maps: int8 [N, R, C] own-side piece maps (create_initial_state inputs, penv:38-60);
        omit both to sample setups on the device."""
        sel = None
        if env_select is not None:
            sel = env_select.to(device=self.device, dtype=torch.uint8).contiguous()
        m1 = m2 = None
        if p1_maps is not None:
            m1 = torch.as_tensor(p1_maps).to(device=self.device, dtype=torch.int8).contiguous().reshape(self.num_envs, -1)
            m2 = torch.as_tensor(p2_maps).to(device=self.device, dtype=torch.int8).contiguous().reshape(self.num_envs, -1)
            assert m1.shape[1] == self.R * self.Cc and m2.shape == m1.shape
            pieces = self.variant.max_pieces_on_board    # the record's capture-event list holds 2 x pieces entries
            if int(torch.maximum((m1 != 0).sum(1).max(), (m2 != 0).sum(1).max())) > pieces:
                raise ValueError("a piece map holds more pieces than the %s variant has (%d per side)" % (self.variant.name, pieces))
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_reset(self._h, _ptr(sel), _ptr(m1), _ptr(m2), self._stream()), self._L)
        self._next_actions_fresh = False
        if self._placement_pending and sel is None:
            # placement='search': once, now that every record holds a game the trial launches can render
            self._placement_pending = False
            if self.has_outputs and not self.compact and self.obs.numel() * 4 > 300e6:
                free, _ = torch.cuda.mem_get_info(self.device)
                try:
                    self.placement_report = self.tune_placement(max_extra_bytes=min(8 << 30, free // 4))
                except RuntimeError as e:          # (no memory for candidates next to another tenant: the plain tensors stay)
                    self.placement_report = {'failed': str(e)}
        return self.observe()

    def observe(self, raw=False, emit_obs=True, emit_mask=True):
        """_get_current_obs for every env (no state change); emit_obs / emit_mask = False skip those outputs."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_observe(self._h, _ptr(self.obs) if emit_obs else None, _ptr(self.fobs) if emit_obs else None,
                                           _ptr(self.mask) if emit_mask else None, _ptr(self.player),
                                           (_lib.STEP_RAW_OBS if raw else 0) | self._mode_flags, self._stream()), self._L)
        return self.obs, self.mask, self.player

    def tune_placement(self, trials=None, max_extra_bytes=8 << 30, wide_extra_bytes=0):
        """tune_placement_once with `max_extra_bytes`; if that budget held no candidate of the fast class (none >= 14 % below the
        slowest one) and `wide_extra_bytes` is larger, a second pass spreads the same number of candidates over that wider budget and is
        kept only if it found something faster (partial-observation buffers that stream past the Infinity Cache only).  Returns the
        first pass's report, with report['wide'] = {'obs': [...], 'used': bool, 'peak_extra_bytes': int} when the second pass ran."""
        rep = self.tune_placement_once(trials, max_extra_bytes)
        t = rep.get('obs') or []
        streams = self.obs.numel() * 4 > 300e6
        if t and streams and wide_extra_bytes > max_extra_bytes and min(t) > 0.86 * max(t) and self.fobs is None:
            first = (self.obs, self.mask, self._outputs_owner, self._outputs, self.placement_peak_extra_bytes)
            rep2 = self.tune_placement_once(trials, wide_extra_bytes)
            t2 = rep2.get('obs') or []
            used = bool(t2) and min(t2) < min(t)
            rep['wide'] = {'obs': t2, 'used': used, 'peak_extra_bytes': self.placement_peak_extra_bytes}
            if not used:
                self.obs, self.mask, self._outputs_owner, self._outputs, self.placement_peak_extra_bytes = first
                self.observe()
        return rep

    def tune_placement_once(self, trials=None, max_extra_bytes=8 << 30):
        """Move the big output tensors (obs, fobs, mask) into library-owned buffers picked by a bounded placement trial
        (sgx_alloc_outputs).

        Measured on MI355X: device memory comes in regions of two kinds; an observation buffer lying inside one region runs at
        that region's rate (about 350 or 380-395 us per launch of 65,536 Barrage games), one whose pages mix both kinds at
        313-325 us, and which of these a plain allocation gets depends on what was allocated before it.  The library tries
        up to `trials` candidate allocations, each behind a padding allocation of growing size that is released again, times a
        few state-preserving sgx_observe launches on each and keeps the fastest; it never holds more than `max_extra_bytes`
        beyond the buffers it returns.  Call after reset().  Returns the per-candidate launch times in microseconds:
        {'obs': [...], ('fobs': [...])}; [0] is the plain first allocation."""
        if self.compact:
            raise ValueError("tune_placement: compact outputs (235 MB per 65,536 Barrage games) fit the Infinity Cache and have no placement classes")
        out = _lib.SgxOutputs()
        flags = self._mode_flags | (_lib.OUT_FULL_OBS if self.fobs is not None else 0)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_alloc_outputs(self._h, flags, int(max_extra_bytes), int(trials or _lib.OUT_MAX_TRIALS),
                                                 self._stream(), C.byref(out)), self._L)
        owner = _OutputsOwner(self._L, out)
        self._outputs_owner = owner          # replaces the env's reference to an earlier allocation (its tensors keep theirs)
        self._outputs = out
        N, R, Cc, K = self.num_envs, self.R, self.Cc, self.K
        self.obs = _wrap_device(out.obs_dev, (N, R, Cc, self.p_channels), torch.float32, self.device, owner)
        self.mask = _wrap_device(out.mask_dev, (N, R, Cc, K), torch.uint8, self.device, owner)
        report = {'obs': [float(x) for x in out.trial_us[:out.n_trials]]}
        if out.fobs_dev:
            self.fobs = _wrap_device(out.fobs_dev, (N, R, Cc, self.f_channels), torch.float32, self.device, owner)
            report['fobs'] = [float(x) for x in out.ftrial_us[:out.n_ftrials]]
        self.placement_peak_extra_bytes = int(out.peak_extra_bytes)
        self.observe()
        return report

    @property
    def record_bytes(self):
        """Bytes of one game's packed state record in device memory (sgx_record_bytes): read once and written once per step."""
        return int(self._L.sgx_record_bytes(self._h))

    @property
    def build_id(self):
        """Hash of the sources the loaded library was compiled from (sgx_build_id)."""
        return self._L.sgx_build_id().decode('ascii', 'replace')

    def alloc_output_ring(self, n_sets, tune=None, max_extra_bytes=8 << 30, trials=None, wide_extra_bytes=0):
        """A ring of `n_sets` output sets (obs, mask[, fobs]) for rollout_steps(..., ring=True): step i writes set i mod n_sets -- a
        rollout into a trajectory buffer that keeps the last n_sets steps (sgx_step_ring).  Set 0 is the env's current set; the others
        are torch.empty tensors or, with tune=True (the default, tune=None, where the env's own set came from the placement search --
        placement='search' or tune_placement()), library-owned buffers from the placement trial (one sgx_alloc_outputs each; a set
        whose search ends more than 3 % above the env's own set is searched once more over `wide_extra_bytes`, like tune_placement's
        wide pass, and the faster of the two is kept; if the extra sets turn out more than 3 % faster than the env's own set, that one
        is searched once more against them).  Returns the per-set trial reports (None for untuned sets and for set 0 unless it was
        searched again: {'obs', 'used', 'target_us', 'before_us'} then)."""
        n_sets = int(n_sets)
        if n_sets < 1:
            raise ValueError("n_sets must be >= 1")
        if tune is None:
            tune = getattr(self, '_outputs', None) is not None and not self.compact
        # every extra set should be as fast as the env's own: the searches go on until a candidate is within 3 % of what the first
        # search kept (sgx_set_placement_target), inside the same budget
        target = 0.0
        if tune and not self.compact and getattr(self, '_outputs', None) is not None and self._outputs.n_trials > 0:
            target = min(float(x) for x in self._outputs.trial_us[:self._outputs.n_trials])
        if not self.has_outputs:
            raise ValueError("alloc_output_ring: this env was created with outputs=False")
        try:      # (whatever a placement call raises midway, the handle's search target goes back to 0: a later tune_placement() must not chase a stale one)
            _lib.check(self._L.sgx_set_placement_target(self._h, C.c_float(target)), self._L)
            N, R, Cc, K, dev = self.num_envs, self.R, self.Cc, self.K, self.device
            self._ring = [(self.obs, self.mask, self.fobs)]
            self._ring_owners = [getattr(self, '_outputs_owner', None)]
            reports = [None]
            for _ in range(1, n_sets):
                if tune and not self.compact:
                    out = _lib.SgxOutputs()
                    flags = self._mode_flags | (_lib.OUT_FULL_OBS if self.fobs is not None else 0)
                    with torch.cuda.device(self.device):
                        _lib.check(self._L.sgx_alloc_outputs(self._h, flags, int(max_extra_bytes), int(trials or _lib.OUT_MAX_TRIALS),
                                                             self._stream(), C.byref(out)), self._L)
                    owner = _OutputsOwner(self._L, out)
                    times = [float(x) for x in out.trial_us[:out.n_trials]]
                    report = {'obs': times}
                    if target > 0 and times and min(times) > 1.03 * target and wide_extra_bytes > max_extra_bytes:
                        out2 = _lib.SgxOutputs()
                        with torch.cuda.device(self.device):
                            _lib.check(self._L.sgx_alloc_outputs(self._h, flags, int(wide_extra_bytes), int(trials or _lib.OUT_MAX_TRIALS),
                                                                 self._stream(), C.byref(out2)), self._L)
                        owner2 = _OutputsOwner(self._L, out2)
                        times2 = [float(x) for x in out2.trial_us[:out2.n_trials]]
                        report['wide'] = {'obs': times2, 'used': bool(times2) and min(times2) < min(times)}
                        if report['wide']['used']:
                            out, owner = out2, owner2          # (the first search's buffers go with their owner)
                            report['obs'] = times + times2     # min() over both = what is kept
                        del out2, owner2
                    obs = _wrap_device(out.obs_dev, (N, R, Cc, self.p_channels), torch.float32, dev, owner)
                    mask = _wrap_device(out.mask_dev, (N, R, Cc, K), torch.uint8, dev, owner)
                    fobs = _wrap_device(out.fobs_dev, (N, R, Cc, self.f_channels), torch.float32, dev, owner) if out.fobs_dev else None
                    reports.append(report)
                    self._ring_owners.append(owner)
                else:
                    obs, mask = torch.empty_like(self.obs), torch.empty_like(self.mask)       # (compact outputs: the same compact shapes)
                    fobs = torch.empty((N, R, Cc, self.f_channels), dtype=torch.float32, device=dev) if self.fobs is not None else None
                    reports.append(None)
                    self._ring_owners.append(None)
                self._ring.append((obs, mask, fobs))
            # The other way round: the env's OWN set may be the slow one (its search ran first, with no target -- one box: 298.5 us after 82
            # candidates, then 274-281 us for the extra sets).  It is searched once more, against the best of the others, and replaced if
            # that finds something faster; reports[0] = {'obs': [...], 'used': bool} then.
            best_extra = min([min(r['obs']) for r in reports[1:] if r and r.get('obs')] or [0.0])
            if target > 0 and 0 < best_extra < 0.97 * target and self.fobs is None:
                _lib.check(self._L.sgx_set_placement_target(self._h, C.c_float(best_extra)), self._L)
                first = (self.obs, self.mask, self._outputs_owner, self._outputs, self.placement_peak_extra_bytes)
                rep0 = self.tune_placement_once(trials, max(int(max_extra_bytes), int(wide_extra_bytes)))
                used = bool(rep0.get('obs')) and min(rep0['obs']) < target
                if not used:
                    self.obs, self.mask, self._outputs_owner, self._outputs, self.placement_peak_extra_bytes = first
                    self.observe()
                self._ring[0] = (self.obs, self.mask, self.fobs)
                self._ring_owners[0] = self._outputs_owner
                reports[0] = {'obs': rep0.get('obs') or [], 'used': used, 'target_us': best_extra, 'before_us': target}
        except BaseException:
            self._ring = self._ring_owners = None      # (no half-built ring)
            raise
        finally:
            _lib.check(self._L.sgx_set_placement_target(self._h, C.c_float(0.0)), self._L)
        self._ring_pos = 1 % n_sets          # set 0 holds the current position's outputs: the next step writes set 1
        self._ring_ios = (_lib.SgxStepIO * n_sets)()
        return reports

    def repeat_output_ring(self, n_entries):
        """The ring of alloc_output_ring() as a ring of `n_entries` entries over the SAME buffers (entry i = set i mod n_sets): what a long
        ring costs on the memory of a short one -- beyond 8 entries sgx_step_ring reads the pointers from a device table (bench.py's trajectory leg
        times exactly this).  The buffers are rewritten every n_sets steps as before."""
        if not self._ring:
            raise ValueError("repeat_output_ring() needs alloc_output_ring() first")
        n_entries, base = int(n_entries), len(self._ring)
        if n_entries < base:
            raise ValueError("repeat_output_ring: fewer entries than the ring has sets")
        pos = (self._ring_pos - 1) % base              # the set holding the current position's outputs keeps that role
        self._ring = [self._ring[i % base] for i in range(n_entries)]
        self._ring_owners = [self._ring_owners[i % base] for i in range(n_entries)]
        self._ring_pos = (pos + 1) % n_entries
        self._ring_ios = (_lib.SgxStepIO * n_entries)()

    def _release_outputs(self):
        """Drops the env's own reference to library-owned output buffers; they are freed when the last tensor viewing them goes."""
        self._outputs_owner = None
        self._outputs = None

    def step(self, actions, want_next_actions=False, emit_obs=True, emit_mask=True, flags=0):
        """One env.step() for every env.  actions: int32 [N] flat (R,C,K) indices in each mover's perspective."""
        a = actions
        if a.dtype != torch.int32 or a.device != self.device or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        assert a.numel() == self.num_envs * (4 if (flags & _lib.STEP_ACTIONS_POSITIONS) else 1)
        io = self._fill_io(a, want_next_actions, emit_obs, emit_mask, flags)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_step(self._h, C.byref(io), self._stream()), self._L)
        self._next_actions_fresh = bool(want_next_actions)
        return self.obs, self.mask, self.reward, self.done, self.player

    def step_sync(self, actions, emit_obs=True, emit_mask=True, flags=0):
        """step() and wait until the outputs are complete (sgx_step_sync): the latency path of a handful of games -- up to 8 games on
        a 4-aligned board of more than 32 cells run one workgroup per game and the call returns as soon as the kernel has published
        its outputs, without a stream synchronisation."""
        a = actions
        if a.dtype != torch.int32 or a.device != self.device or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        assert a.numel() == self.num_envs * (4 if (flags & _lib.STEP_ACTIONS_POSITIONS) else 1)
        io = self._fill_io(a, False, emit_obs, emit_mask, flags)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_step_sync(self._h, C.byref(io), self._stream()), self._L)
        self._next_actions_fresh = False
        return self.obs, self.mask, self.reward, self.done, self.player

    def _fill_io(self, a, want_next_actions, emit_obs, emit_mask, flags):
        io = self._io
        io.actions_dev = a.data_ptr()
        io.obs_dev = self.obs.data_ptr() if (emit_obs and self.obs is not None) else None
        io.fobs_dev = self.fobs.data_ptr() if (emit_obs and self.fobs is not None) else None
        io.final_fobs_dev = self.final_fobs.data_ptr() if self.final_fobs is not None else None
        io.mask_dev = self.mask.data_ptr() if (emit_mask and self.mask is not None) else None
        io.reward_dev = self.reward.data_ptr()
        io.done_dev = self.done.data_ptr()
        io.player_dev = self.player.data_ptr()
        io.invalid_action_dev = self.invalid_action.data_ptr()
        io.ending_invalid_dev = self.ending_invalid.data_ptr()
        io.final_obs_dev = self.final_obs.data_ptr() if self.final_obs is not None else None
        io.next_actions_dev = self.next_actions.data_ptr() if want_next_actions else None
        io.auto_reset = 1 if self.auto_reset else 0
        io.flags = int(flags) | self._mode_flags
        return io

    def rollout_step(self):
        """Random-valid-action rollout step: plays `next_actions` (drawn by the previous call, or now if there is no
        current draw) and draws the next."""
        if not self._next_actions_fresh:
            self.sample_valid_actions()
        return self.step(self.next_actions, want_next_actions=True)

    def rollout_steps(self, n_steps, chains=1, ring=False, emit_obs=True, emit_mask=True):
        """`n_steps` rollout steps enqueued by one library call (sgx_step_n): same results as calling rollout_step()
        n_steps times, without a Python round trip per step (toy boards are launch-bound otherwise).  chains > 1 (sgx_rollout):
        the batch is split into that many contiguous ranges of games whose launches overlap on streams of their own; chains = 0 /
        'auto': the multi-step launch where the call is eligible, else as many chains as the library's measured rule says (2 on boards of
        up to 36 cells and on odd boards, else 1).
        ring=True (after alloc_output_ring): the steps write the ring's output sets in turn (sgx_step_ring); self.obs / self.mask /
        self.fobs are the set the LAST step wrote afterwards.  emit_obs / emit_mask = False: rollouts that write no observation / no mask
        (logic-only playouts, e.g. of a search).  Where the call is eligible all steps run in ONE launch (sgx_set_multi_step): the games stay on
        the chip between the steps."""
        if not self._next_actions_fresh:
            self.sample_valid_actions()
        if ring:
            if not self._ring:
                raise ValueError("rollout_steps(ring=True) needs alloc_output_ring() first")
            n_sets, n_steps = len(self._ring), int(n_steps)
            for k, (obs, mask, fobs) in enumerate(self._ring):
                self.obs, self.mask, self.fobs = obs, mask, fobs
                io = self._fill_io(self.next_actions, True, emit_obs, emit_mask, 0)
                C.memmove(C.byref(self._ring_ios[k]), C.byref(io), C.sizeof(_lib.SgxStepIO))
            first = self._ring_pos
            with torch.cuda.device(self.device):
                _lib.check(self._L.sgx_step_ring(self._h, self._ring_ios, n_sets, first, n_steps, self._stream()), self._L)
            last = (first + n_steps - 1) % n_sets if n_steps > 0 else (first - 1) % n_sets
            self.obs, self.mask, self.fobs = self._ring[last]
            self._ring_pos = (first + n_steps) % n_sets
            return self.obs, self.mask, self.reward, self.done, self.player
        io = self._fill_io(self.next_actions, True, emit_obs, emit_mask, 0)
        with torch.cuda.device(self.device):
            if chains in (0, 'auto') or chains > 1:        # 0 / 'auto': the library's measured rule (2 chains on small and odd boards)
                _lib.check(self._L.sgx_rollout(self._h, C.byref(io), int(n_steps), 0 if chains == 'auto' else int(chains), self._stream()), self._L)
            else:
                _lib.check(self._L.sgx_step_n(self._h, C.byref(io), int(n_steps), self._stream()), self._L)
        return self.obs, self.mask, self.reward, self.done, self.player

    # ---- trajectory buffers: a fresh slice per step, like the reference's fresh arrays per env.step() (impl:905, maenv:447-497) ------
    def alloc_trajectory(self, n_slots, results=True, actions=True):
        """Tensors with a leading slot axis for rollout_trajectory(): obs [T,N,R,C,67], mask [T,N,R,C,K] (fobs [T,N,R,C,79] with
        full_obs=True; the compact shapes with compact_outputs=True), and -- results=True -- reward [T,N,2], done / invalid_action /
        ending_invalid [T,N], player [T,N]; actions=True: the action every env DREW in the step, int32 [T,N] (the action the next step
        plays: chosen from the observation and mask of the same slot).  Returns a dict of torch tensors."""
        if not self.has_outputs:
            raise ValueError("alloc_trajectory: this env was created with outputs=False")
        T, N, dev = int(n_slots), self.num_envs, self.device
        if T < 1:
            raise ValueError("n_slots must be >= 1")
        traj = {'obs': torch.empty((T,) + tuple(self.obs.shape), dtype=self.obs.dtype, device=dev),
                'mask': torch.empty((T,) + tuple(self.mask.shape), dtype=self.mask.dtype, device=dev)}
        if self.fobs is not None:
            traj['fobs'] = torch.empty((T,) + tuple(self.fobs.shape), dtype=torch.float32, device=dev)
        if results:
            traj['reward'] = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
            for k, dt in (('done', torch.uint8), ('invalid_action', torch.uint8), ('ending_invalid', torch.uint8), ('player', torch.int8)):
                traj[k] = torch.zeros((T, N), dtype=dt, device=dev)
        if actions:
            traj['actions'] = torch.zeros((T, N), dtype=torch.int32, device=dev)
        return traj

    def rollout_trajectory(self, n_steps, traj, first_slot=0, emit_obs=True, emit_mask=True):
        """`n_steps` random-valid-action rollout steps (like rollout_steps) whose step t writes slot (first_slot + t) % T of the
        tensors of `traj` (alloc_trajectory): one library call (sgx_step_traj), and where the call is eligible ONE launch per 256 steps
        (sgx_set_multi_step) -- for any number of slots.  Afterwards self.obs / self.mask / self.fobs (and, if `traj` holds per-slot
        results, self.reward / done / player / invalid_action / ending_invalid) are views of the slot the LAST step wrote; next_actions
        holds the last draw.  Returns `traj`."""
        n_steps = int(n_steps)
        T = int(traj['obs'].shape[0])
        if not 0 <= first_slot < T:
            raise ValueError("first_slot out of range")
        for k in ('obs', 'mask'):
            if tuple(traj[k].shape[1:]) != tuple(getattr(self, k).shape) or traj[k].dtype != getattr(self, k).dtype or not traj[k].is_contiguous():
                raise ValueError("traj['%s'] must be a contiguous [T, %s] %s tensor" % (k, ', '.join(map(str, getattr(self, k).shape)), getattr(self, k).dtype))
        if (self.fobs is not None) != ('fobs' in traj):
            raise ValueError("traj['fobs'] goes with full_obs=True")
        per_slot = 'reward' in traj
        if not self._next_actions_fresh:
            self.sample_valid_actions()
        keep = (self.obs, self.mask, self.fobs, self.reward, self.done, self.player, self.invalid_action, self.ending_invalid)

        def views(slot):
            self.obs, self.mask, self.fobs = traj['obs'][slot], traj['mask'][slot], (traj['fobs'][slot] if 'fobs' in traj else None)
            if per_slot:
                self.reward, self.done, self.player = traj['reward'][slot], traj['done'][slot], traj['player'][slot]
                self.invalid_action, self.ending_invalid = traj['invalid_action'][slot], traj['ending_invalid'][slot]
        try:
            views(0)
            io = self._fill_io(self.next_actions, True, emit_obs, emit_mask, 0)
            t = _lib.SgxTrajIO()
            C.memmove(C.byref(t.io), C.byref(io), C.sizeof(_lib.SgxStepIO))
            t.n_slots, t.results_per_slot, t.slot_envs = T, 1 if per_slot else 0, self.num_envs
            t.actions_log_dev = traj['actions'].data_ptr() if 'actions' in traj else None
            with torch.cuda.device(self.device):
                _lib.check(self._L.sgx_step_traj(self._h, C.byref(t), int(first_slot), n_steps, self._stream()), self._L)
        except BaseException:
            (self.obs, self.mask, self.fobs, self.reward, self.done, self.player, self.invalid_action, self.ending_invalid) = keep
            raise
        if n_steps == 0:
            (self.obs, self.mask, self.fobs, self.reward, self.done, self.player, self.invalid_action, self.ending_invalid) = keep
        else:
            views((first_slot + n_steps - 1) % T)
        return traj

    def choose_actions(self, logits, temperature=1.0, mask=None, out=None):
        """The reference's chooser for a batch (examples/basic_game_loop.py:6-31): one action per game drawn from
        softmax(logits / temperature) over the VALID actions of `mask` (default: the current mask -- bytes, or the bit mask of
        compact_outputs=True), on the device (sgx_choose_actions).  logits: float32 [N, R, C, K] (or [N, R*C*K]) from the caller's
        policy.  temperature 0 = argmax.  The draw is keyed by (seed, global env id, game, turn) like the fused sampler's, and with equal
        logits it is that sampler's action.  Weights are 2^23 fixed point: an action less than 2^-23 as likely as the most likely one is
        never drawn; an env whose valid actions ALL have -inf / NaN logits (or whose mask is empty) gets -1, which step() would flag as
        an invalid action.  -> int32 [N] (`out` or next_actions: ready for step())."""
        lg = logits
        if lg.dtype != torch.float32 or lg.device != self.device or not lg.is_contiguous():
            lg = lg.to(device=self.device, dtype=torch.float32).contiguous()
        if lg.numel() != self.num_envs * self.R * self.Cc * self.K:
            raise ValueError("logits must hold num_envs x rows x columns x ways_to_move floats")
        own = mask is None
        mask = self.mask if mask is None else mask
        bits = mask.dtype == torch.int32
        if not bits and mask.numel() != lg.numel():
            raise ValueError("mask must be uint8 [N, R, C, K] (or the int32 bit mask of a compact step)")
        out = self.next_actions if out is None else out
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_choose_actions(self._h, _ptr(lg), _ptr(mask), C.c_float(float(temperature)),
                                                  _lib.STEP_COMPACT_MASK if bits else 0, _ptr(out), self._stream()), self._L)
        if own and out is self.next_actions:
            self._next_actions_fresh = False      # (a policy's choice, not the uniform draw rollout_step() plays)
        return out

    # ---- snapshots: the packed records of every game, in a pool without output tensors -------------------------------
    def snapshot(self, out=None):
        """Copy every game's packed record (boards, counters, game number: everything the counter RNG and the rules depend on)
        into a pool of records without output tensors (`out`, or a new one): sgx_copy_envs.  restore() puts it back."""
        snap = out if out is not None else VecStrategoEnv(self.variant, self.num_envs, device=self.device.index, seed=self.seed,
                                                          env_id_offset=self.env_id_offset, human_inits=False, outputs=False)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_copy_envs(snap._h, None, self._h, None, self.num_envs, self._stream()), self._L)
        return snap

    def restore(self, snap):
        """The inverse of snapshot(): every game continues from the snapshot's position; returns observe() of it (the outputs and the
        next random draw are those of the restored position again)."""
        if snap.num_envs != self.num_envs:
            raise ValueError("restore: the snapshot holds %d games, this env %d" % (snap.num_envs, self.num_envs))
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_copy_envs(self._h, None, snap._h, None, self.num_envs, self._stream()), self._L)
        self._next_actions_fresh = False
        return self.observe()

    def sample_valid_actions(self, mask=None, out=None):
        """Uniformly random valid action per env from `mask` (default: the current one) -- maenv:830-834."""
        own = mask is None or mask is self.mask
        mask = self.mask if mask is None else mask
        out = self.next_actions if out is None else out
        src = mask
        if own and self.compact:                 # the standalone sampler reads mask BYTES: expand the bit mask first
            self._mask_bytes = self.decode_mask(self._mask_bytes)
            src = self._mask_bytes
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_sample_valid(self._h, _ptr(src), _ptr(out), self._stream()), self._L)
        if own and out is self.next_actions:
            self._next_actions_fresh = True
        return out

    def export_state(self):
        """(state int64 [N,34,R,C] in the reference layout, player int8 [N])."""
        st = torch.empty((self.num_envs, NUM_STATE_LAYERS, self.R, self.Cc), dtype=torch.int64, device=self.device)
        pl = torch.empty((self.num_envs,), dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_export_state(self._h, _ptr(st), _ptr(pl), self._stream()), self._L)
        return st, pl

    def import_state(self, state, player=None):
        st = torch.as_tensor(state).to(device=self.device, dtype=torch.int64).contiguous()
        assert tuple(st.shape) == (self.num_envs, NUM_STATE_LAYERS, self.R, self.Cc)
        pl = None
        if player is not None:
            pl = torch.as_tensor(player).to(device=self.device, dtype=torch.int8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_import_state(self._h, _ptr(st), _ptr(pl), self._stream()), self._L)
        self._next_actions_fresh = False
        return self.observe()

    def import_state_checked(self, state, player, sanitised):
        """import_state without the observe, reporting altered states in `sanitised` (uint8 [N], see sgx_import_state_checked)."""
        st = torch.as_tensor(state).to(device=self.device, dtype=torch.int64).contiguous()
        assert tuple(st.shape) == (self.num_envs, NUM_STATE_LAYERS, self.R, self.Cc)
        pl = None if player is None else torch.as_tensor(player).to(device=self.device, dtype=torch.int8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_import_state_checked(self._h, _ptr(st), _ptr(pl), _ptr(sanitised), self._stream()), self._L)
        self._next_actions_fresh = False

    def env_info(self):
        """int32 [N,4]: turn count, game number, game_over, current player."""
        out = torch.empty((self.num_envs, 4), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sgx_get_env_info(self._h, _ptr(out), self._stream()), self._L)
        return out
