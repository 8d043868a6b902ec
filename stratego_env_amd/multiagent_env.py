"""StrategoMultiAgentEnv: the reference's 2-agent dict API (stratego_multiagent_env.py:316-834) on the HIP path.

Drop-in for examples/basic_game_loop.py: same constructor config keys, `reset()` / `step(action_dict)` return
the same dict structures with the same keys (ObservationComponents string values, players +1 / -1,
"__all__"), the same dtypes (mask int64 (R,C,K), partial observation float32 (R,C,67)), the same error
behaviour (ValueError on an invalid action, AssertionError when the wrong player acts) and the same use of
numpy's / Python's global RNGs at reset, so `np.random.seed(s); random.seed(s)` reproduces the reference's
setups.  Every game-logic operation runs in the HIP kernels through a VecStrategoEnv of one game.

All three observation modes (PARTIALLY_OBSERVABLE, FULLY_OBSERVABLE, BOTH_OBSERVATIONS = the reference default), both channel
modes ('extended' 67 / 79 channels, 'original' 32 / 33) and curriculum start states (`curriculum_start_states_path`: `.npz`, or the
reference's HDF5 file where h5py is installed) are built.  Not built (out of the hot-path scope): the vs_human GUI and
the vs_bot socket link.
"""
import copy

import numpy as np
import torch

from . import _lib, index_algebra as ia, obs_norm
from .config import FO_OBS_CHANNELS, PO_OBS_CHANNELS, NUM_STATE_LAYERS, get_variant
from .enums import GameVersions, ObservationComponents, ObservationModes
from .setups import load_setup_table, sample_initial_maps_like_reference
from .spaces import Box, Dict, Discrete
from .procedural_env import StrategoProceduralEnv
from .vec_env import VecStrategoEnv

SPATIAL_STRATEGO_ENV = 'SpatialStratego-v1'   # maenv:31

_NP_DTYPES = {torch.uint8: np.uint8, torch.int8: np.int8, torch.int32: np.int32, torch.float32: np.float32}

DEFAULT_CONFIG = {   # maenv:47-69
    'version': GameVersions.STANDARD,
    'repeat_games_from_other_side': False,
    'random_player_assignment': False,
    'observation_mode': ObservationModes.BOTH_OBSERVATIONS,
    'observation_includes_internal_state': False,
    'vs_bot': False,
    'bot_player_num': 1,
    'fixed_bot_player_num': True,
    'bot_relative_path': 'basic_python.py',
    'vs_human': False,
    'human_player_num': -1,
    'human_web_gui_port': 7000,
    'human_inits': False,
    'penalize_ties': False,
    'curriculum_start_states_path': None,
    'obs_channel_mode': 'extended',
    'same_start_pos_everytime': False,
}

_MASK = ObservationComponents.VALID_ACTIONS_MASK.value
_POBS = ObservationComponents.PARTIAL_OBSERVATION.value
_FOBS = ObservationComponents.FULL_OBSERVATION.value
_ISTATE = ObservationComponents.INTERNAL_STATE.value

# layer pairs swapped by the perspective flip (impl:645-675)
_SWAP = [(0, 1), (3, 4), (6, 7), (32, 33)] + [(8 + k, 20 + k) for k in range(12)]


def state_from_player_perspective(state, player):
    """impl:645-675 on a host copy of the int64 [34,R,C] state (used for INTERNAL_STATE and side-swapped replays)."""
    if player == 1:
        return state
    out = state.copy()
    for a, b in _SWAP:
        out[a] = state[b, ::-1, ::-1]
        out[b] = state[a, ::-1, ::-1]
    out[2] = state[2, ::-1, ::-1]
    return out


def load_curriculum_start_states(path):
    """Curriculum start states (util.py:322-387): datasets 'state' [n,34,R,C] and 'winner' [n] (the likely winner, +1/-1).

    `.npz` files are read with numpy; anything else is opened as HDF5 like the reference does (`h5py.File`, util.py:327) -- with h5py
    where it is installed, else with the package's own small reader (stratego_env_amd/hdf5_lite.py: the MI355X image has no h5py; it
    reads what h5py writes for plain numeric datasets -- contiguous or chunked, gzip / shuffle / fletcher32, old and new file
    format -- and is tested against files written by the real h5py and against what the reference reads from them,
    tests/test_hdf5_lite.py).  The reference re-opens the file on every reset; the table is loaded once here (same draws from `np.random`)."""
    if str(path).endswith('.npz'):
        with np.load(path) as z:
            return np.asarray(z['state']), np.asarray(z['winner'])
    try:
        import h5py
    except ImportError:
        from . import hdf5_lite
        with hdf5_lite.File(path) as f:
            return np.asarray(f['state']), np.asarray(f['winner'])
    with h5py.File(path, 'r') as f:
        return np.asarray(f['state']), np.asarray(f['winner'])


def resolve_variant(env_config, human_inits=None):
    """(variant of the games, variant of the SETUPS) for an env_config.

    The reference merges env_config OVER the version's config dict (maenv:320-323, with_base_config maenv:72-77), so a caller may
    override single variant fields -- and only some code paths honour them (reproduced; tests/golden/facade_overrides.json):
      rows / columns / piece_amounts   the merged values: board of the operator object and observation normalisation (maenv:325-329,
                                       370-382); the SETUPS keep the version's pieces (random: VERSION_CONFIGS, maenv:347-349;
                                       human: the Gravon strings), so piece_amounts changes the normalisation only;
      max_turns / obstacle_locations   honoured with human_inits (get_random_human_init_fn gets the merged dict, maenv:336-338);
                                       the random and the curriculum paths read VERSION_CONFIGS[version] (maenv:340-349);
      initial_state_usable_rows        read by nothing but the random path, i.e. from VERSION_CONFIGS: no effect."""
    user = env_config if env_config else {}
    base = get_variant(user.get('version', DEFAULT_CONFIG['version']))
    if human_inits is None:
        human_inits = user.get('human_inits', DEFAULT_CONFIG['human_inits'])
    ref_cfg = base.as_reference_config()
    overridden = [k for k in ('rows', 'columns', 'max_turns', 'obstacle_locations', 'piece_amounts', 'initial_state_usable_rows')
                  if k in user and user[k] != ref_cfg[k]]
    if not overridden:
        return base, base
    from .config import custom_variant, piece_counts_from_amounts
    merged = {k: (user[k] if k in user else ref_cfg[k]) for k in ref_cfg}
    init_src = merged if human_inits else ref_cfg
    counts = piece_counts_from_amounts(merged['piece_amounts'])
    rows, columns = int(merged['rows']), int(merged['columns'])
    obstacles = [tuple(x) for x in init_src['obstacle_locations'] if x[0] < rows and x[1] < columns]
    variant = custom_variant(rows, columns, max_turns=init_src['max_turns'], obstacle_locations=obstacles, piece_counts=counts,
                             initial_state_usable_rows=min(base.initial_state_usable_rows, max(1, rows // 2)),
                             name='%s_overridden_%s' % (base.name, '_'.join(overridden)), human_inits=base.human_inits,
                             capture_capacity=min(max(sum(counts), base.pieces_per_side), rows * columns // 2))
    return variant, base


class StrategoMultiAgentEnv:

    def __init__(self, env_config=None, device=0):
        cfg = copy.deepcopy(DEFAULT_CONFIG)
        cfg.update(env_config if env_config else {})
        # (variant the kernels run, variant the setup functions see): differ when env_config overrides fields of the version's config
        self.variant, self._setup_variant = resolve_variant(env_config, cfg['human_inits'])
        v = self.variant
        for key in ('vs_human', 'vs_bot'):
            if cfg[key]:
                raise NotImplementedError("%s is outside the MI355X hot-path build (SURVEY.md section 8)" % key)
        if cfg['obs_channel_mode'] not in ('extended', 'original'):
            raise ValueError("obs_channel_mode must be 'extended' or 'original'")
        self._extended_channels = cfg['obs_channel_mode'] == 'extended'                           # maenv:368
        mode = cfg['observation_mode']
        if isinstance(mode, str):
            mode = ObservationModes(mode)
        assert mode in (ObservationModes.PARTIALLY_OBSERVABLE, ObservationModes.FULLY_OBSERVABLE,
                        ObservationModes.BOTH_OBSERVATIONS)                                        # maenv:384-385
        self.observation_mode = mode
        self._want_p = mode in (ObservationModes.PARTIALLY_OBSERVABLE, ObservationModes.BOTH_OBSERVATIONS)
        self._want_f = mode in (ObservationModes.FULLY_OBSERVABLE, ObservationModes.BOTH_OBSERVATIONS)
        self.vs_human = self.vs_bot = False                                                 # maenv:429, 436 (GUI / bot links are not built)
        self.penalize_ties = cfg['penalize_ties']
        self.random_player_assignment = cfg['random_player_assignment']
        assert not (cfg['human_inits'] and cfg['curriculum_start_states_path'])           # maenv:332
        self.use_curriculum_inits = False
        if cfg['curriculum_start_states_path'] and not cfg['human_inits']:                  # maenv:341-346
            self.use_curriculum_inits = True
            self.random_player_assignment = True
            self._curriculum = load_curriculum_start_states(cfg['curriculum_start_states_path'])
        self.repeat_games_from_other_side = cfg['repeat_games_from_other_side']
        assert not (self.random_player_assignment and self.repeat_games_from_other_side)   # maenv:358
        self.observation_includes_internal_state = cfg['observation_includes_internal_state']
        self.human_inits = bool(cfg['human_inits'])
        if self.human_inits and not v.human_inits:
            raise ValueError("Human inits not supported with {} game version".format(v.name))   # util.py:310
        self._table = load_setup_table(v.human_inits) if self.human_inits else None

        self._vec = VecStrategoEnv(v, 1, device=device, seed=0, human_inits=False, auto_reset=False, final_obs=True,
                                   full_obs=self._want_f, obs_channel_mode=cfg['obs_channel_mode'])
        self._p_obs_num_layers, self._f_obs_num_layers = self._vec.p_channels, self._vec.f_channels   # maenv:370-382
        original = not self._extended_channels
        self._p_obs_highs, self._p_obs_lows = obs_norm.obs_highs_lows(v.piece_counts, full=False, original=original)
        self._f_obs_highs, self._f_obs_lows = obs_norm.obs_highs_lows(v.piece_counts, full=True, original=original)
        self._p_obs_ranges, self._p_obs_mids = obs_norm.ranges_mids(self._p_obs_highs, self._p_obs_lows)   # maenv:388-391
        self._f_obs_ranges, self._f_obs_mids = obs_norm.ranges_mids(self._f_obs_highs, self._f_obs_lows)   # maenv:393-396
        self._build_host_mirror()
        # the operator-level object the reference exposes (maenv:329); its one-game handle is created on first use
        self.base_env = StrategoProceduralEnv(v.rows, v.columns, version=v, device=device)
        self.rows, self.columns = v.rows, v.columns
        self.spatial_action_size = v.spatial_action_size
        self.action_size = v.action_size

        self._fixed_maps = self._fixed_curriculum = None
        if cfg['same_start_pos_everytime']:                       # maenv:352-354
            if self.use_curriculum_inits:
                self._fixed_curriculum = self._draw_curriculum_start()
            else:
                self._fixed_maps = self._random_initial_maps()

        self.episodes_completed = 0
        self.last_initial_state = None
        self.action_space = Discrete(int(np.prod(self.spatial_action_size)))          # maenv:362
        spaces = {_MASK: Box(np.float32(0), np.float32(1), self.spatial_action_size)}             # maenv:398-417
        if self._want_p:
            spaces[_POBS] = Box(np.float32(-1.0), np.float32(1.0), (v.rows, v.columns, self._p_obs_num_layers))
        if self._want_f:
            spaces[_FOBS] = Box(np.float32(-1.0), np.float32(1.0), (v.rows, v.columns, self._f_obs_num_layers))
        if self.observation_includes_internal_state:
            if not self.use_curriculum_inits:
                self._random_initial_maps()      # the reference samples a state here just for its shape (maenv:419-425): same draws
            spaces[_ISTATE] = Box(np.float32(-np.inf), np.float32(np.inf), (NUM_STATE_LAYERS, v.rows, v.columns))
        self.observation_space = Dict(spaces)
        self.player = 1
        self.player_map = lambda p: p
        self.reverse_player_map = lambda p: p

    # ---- maenv:499-511 (the kernels emit normalised observations; these are the reference's public helpers) -----
    def normalize_f_observation(self, f_obs):
        return (f_obs - self._f_obs_mids) / self._f_obs_ranges

    def denormalize_f_observation(self, f_obs):
        return (f_obs * self._f_obs_ranges) + self._f_obs_mids

    def normalize_p_observation(self, p_obs):
        return (p_obs - self._p_obs_mids) / self._p_obs_ranges

    def denormalize_p_observation(self, p_obs):
        return (p_obs * self._p_obs_ranges) + self._p_obs_mids

    # ---- setup sampling with the reference's RNG consumption -------------------------------------------
    def _random_initial_maps(self):
        v = self._setup_variant          # the VERSION's config: the reference's setup functions never see overridden piece_amounts /
        if self._fixed_maps is not None:  # usable rows (maenv:347-349, util.py:241-275)
            return self._fixed_maps
        if (v.rows, v.columns) != (self.variant.rows, self.variant.columns):
            # the reference builds the version's rows x columns piece maps and create_initial_state refuses them (penv:44-55)
            raise ValueError("player_1_initial_piece_map (shape {}) is not the correct shape. (Should be {})".format(
                (v.rows, v.columns), (self.variant.rows, self.variant.columns)))
        return sample_initial_maps_like_reference(v, self._table)

    # ---- state access -------------------------------------------------------------------------------------
    @property
    def state(self):
        """int64 [34,R,C] in the reference layout, absolute coordinates (a host copy)."""
        st, _ = self._vec.export_state()
        return st[0].cpu().numpy()

    def _build_host_mirror(self):
        """Single-game latency: all per-step outputs of the one-env batch live in ONE device slab with a pinned host
        mirror, so a step is one 4-byte upload, one kernel launch, one download and one synchronisation.  Order: the
        small flags, rewards, mask, observations, then the terminal observations (fetched on terminal steps only)."""
        vec = self._vec
        names = ['invalid_action', 'done', 'player', 'ending_invalid', 'reward', 'mask', 'obs', 'fobs', 'final_obs', 'final_fobs']
        layout, off = [], 0
        for n in names:
            t = getattr(vec, n)
            if t is None:
                continue
            nbytes = t.numel() * t.element_size()
            layout.append((n, off, nbytes, t.dtype, tuple(t.shape)))
            off = (off + nbytes + 255) & ~255
            if n in ('obs', 'fobs'):
                self._step_bytes = off                      # everything a non-terminal step returns
        self._slab = torch.zeros(off, dtype=torch.uint8, device=vec.device)
        self._host = torch.zeros(off, dtype=torch.uint8).pin_memory()
        self._host_np = self._host.numpy()
        self._view = {}
        for n, o, nbytes, dtype, shape in layout:
            setattr(vec, n, self._slab[o:o + nbytes].view(dtype).view(shape))
            self._view[n] = self._host_np[o:o + nbytes].view(_NP_DTYPES[dtype]).reshape(shape)
        # step(): the same layout once more in pinned host memory that the device addresses (sgx_host_alloc), plus the action word:
        # the kernel reads the action from there and writes the game's outputs there, and env.step() is ONE library call
        # (sgx_step_sync) -- no upload, no download, no torch call on the path (19 k -> ~40 k steps/s, tools/facade_latency.py)
        import ctypes as C
        hp, dp = C.c_void_p(), C.c_void_p()
        _lib.check(vec._L.sgx_host_alloc(vec._h, off + 256, C.byref(hp), C.byref(dp)), vec._L)
        self._hslab_host, self._hslab_dev = hp.value, dp.value
        raw = np.ctypeslib.as_array((C.c_uint8 * (off + 256)).from_address(hp.value))
        self._hview = {n: raw[o:o + nbytes].view(_NP_DTYPES[dtype]).reshape(shape) for n, o, nbytes, dtype, shape in layout}
        self._act_word = raw[off:off + 4].view(np.int32)
        # the four byte flags of a step, read as Python ints straight from the slab (a numpy scalar per flag costs 4 x 0.15 us)
        self._hbytes = memoryview(raw)
        offs = {n: o for n, o, nbytes, dtype, shape in layout}
        self._flag_offs = (offs['invalid_action'], offs['done'], offs['player'], offs['ending_invalid'])
        self._step_stream = vec._stream()                 # (refreshed by reset(): the caller's current stream at that time)
        io = _lib.SgxStepIO()
        dev = {n: dp.value + o for n, o, nbytes, dtype, shape in layout}
        io.actions_dev = dp.value + off
        io.obs_dev, io.fobs_dev, io.mask_dev = dev.get('obs'), dev.get('fobs'), dev['mask']
        io.reward_dev, io.done_dev, io.player_dev = dev['reward'], dev['done'], dev['player']
        io.invalid_action_dev, io.ending_invalid_dev = dev['invalid_action'], dev['ending_invalid']
        io.final_obs_dev, io.final_fobs_dev = dev.get('final_obs'), dev.get('final_fobs')
        io.next_actions_dev, io.auto_reset = None, 0
        self._step_io = io

    def _fetch(self, nbytes):
        self._host[:nbytes].copy_(self._slab[:nbytes], non_blocking=True)
        torch.cuda.current_stream(self._vec.device).synchronize()

    def _obs_dict(self, obs_h, fobs_h, mask_h, player):
        d = {_MASK: mask_h.astype(np.int64)}
        if self._want_p:
            d[_POBS] = obs_h.copy()
        if self._want_f:
            d[_FOBS] = fobs_h.copy()
        if self.observation_includes_internal_state:
            d[_ISTATE] = state_from_player_perspective(self.state, player)             # maenv:494-495
        return d

    def _draw_curriculum_start(self):
        """random_human_init of get_random_curriculum_init_fn (util.py:372-387): one np.random.randint draw."""
        states, winners = self._curriculum
        offset = np.random.randint(low=0, high=len(states))
        st = np.squeeze(np.asarray(states[offset])).astype(np.int64)
        st[5, 0, 0] = 0                                                                 # StateData.TURN_COUNT
        st[5, 1, 0] = self.variant.max_turns                                            # StateData.MAX_TURNS
        return st, int(np.squeeze(winners[offset]))

    # ---- reference API ----------------------------------------------------------------------------------------
    def reset(self, first_player_override=None, initial_state_override=None):
        v = self.variant
        if self.use_curriculum_inits:                                                   # maenv:519-527, util.py:372-387
            st, likely_winner = self._fixed_curriculum if self._fixed_curriculum is not None else self._draw_curriculum_start()
            self.player = int(np.random.choice([-1, 1]))
            self._vec.import_state(st[None], np.asarray([self.player], dtype=np.int8))
            # player 1 gets the advantage of the curriculum start
            self.player_map = lambda p: likely_winner if p == 1 else (-likely_winner if p == -1 else p)
            self.reverse_player_map = lambda p: 1 if p == likely_winner else (-1 if p == -likely_winner else p)
            return self._finish_reset(first_player_override, initial_state_override)
        if self.repeat_games_from_other_side and self.episodes_completed % 2 == 1:      # maenv:530-534
            initial_state = state_from_player_perspective(self.last_initial_state, -1)
            self._vec.import_state(initial_state[None], np.asarray([-1], dtype=np.int8))
            self.player = -1
        else:
            if self.random_player_assignment:                                          # maenv:537-543
                if np.random.random() < 0.5:
                    self.player_map = lambda p: p
                    self.reverse_player_map = lambda p: p
                else:
                    self.player_map = lambda p: -p if p != "__all__" else p
                    self.reverse_player_map = lambda p: -p if p != "__all__" else p
            m1, m2 = self._random_initial_maps()                                       # maenv:545
            self._vec.reset(np.asarray(m1, dtype=np.int8)[None], np.asarray(m2, dtype=np.int8)[None])
            self.player = 1
            initial_state = self.state
        self.last_initial_state = initial_state
        return self._finish_reset(first_player_override, initial_state_override)

    def _finish_reset(self, first_player_override, initial_state_override):
        v = self.variant
        if initial_state_override is not None:                                         # maenv:551-553
            st = np.asarray(initial_state_override, dtype=np.int64)
            if st.shape != (NUM_STATE_LAYERS, v.rows, v.columns):
                raise ValueError("initial_state_override must have shape (34, rows, columns)")
            self._vec.import_state(st[None], np.asarray([self.player], dtype=np.int8))
        if first_player_override is not None:                                          # maenv:555-558
            if not (first_player_override == 1 or first_player_override == -1):
                raise ValueError("first_player_override must either be 1 or -1 if it is not set to None.")
            self.player = int(first_player_override)
            self._vec.import_state(self.state[None], np.asarray([self.player], dtype=np.int8))
        self.episodes_completed += 1
        self._step_stream = self._vec._stream()
        self._vec.observe()
        self._fetch(self._step_bytes)
        hv = self._view
        obs = {self.player: self._obs_dict(hv['obs'][0], hv['fobs'][0] if self._want_f else None, hv['mask'][0], self.player)}
        if self.random_player_assignment:                                              # maenv:654-655
            obs = {self.player_map(k): val for k, val in obs.items()}
        return obs

    def step(self, action_dict, check_for_human_move=True, check_for_bot_move=True, allow_piece_oscillation=False,
             is_spatial_index=True):
        if self.random_player_assignment:                                              # maenv:674-675
            action_dict = {self.reverse_player_map(k): val for k, val in action_dict.items()}
        assert self.player in action_dict                                              # maenv:678-679
        assert self.player * -1 not in action_dict
        action = int(action_dict[self.player])
        vec = self._vec
        flags = _lib.STEP_ALLOW_OSCILLATION if allow_piece_oscillation else 0
        if not is_spatial_index:
            # a 1-D index in the mover's perspective (maenv:684-689): flip to absolute coordinates on the host
            action = int(ia.action_1d_from_player_perspective(self.rows, self.columns, action, self.player))
            flags |= _lib.STEP_ACTIONS_1D
        if not (-2 ** 31 <= action < 2 ** 31):                                         # np.unravel_index raises (maenv:685)
            raise ValueError("Couldn't get the next state because the move wasn't valid.")
        self._act_word[0] = action
        io = self._step_io
        io.flags = flags | vec._mode_flags
        rc = vec._L.sgx_step_sync(vec._h, io, self._step_stream)
        if rc:
            _lib.check(rc, vec._L)
        vec._next_actions_fresh = False
        hv, hb, fo = self._hview, self._hbytes, self._flag_offs
        flags = (hb[fo[0]], hb[fo[1]], 1 if hb[fo[2]] == 1 else -1, hb[fo[3]])
        if flags[0]:
            raise ValueError("Couldn't get the next state because the move wasn't valid.")   # impl:902
        self.player = flags[2]
        if not flags[1]:                                                                # maenv:767-770
            dones = {self.player: False, "__all__": False}
            obs = {self.player: self._obs_dict(hv['obs'][0], hv['fobs'][0] if self._want_f else None, hv['mask'][0], self.player)}
            rewards = {self.player: 0}
            infos = {}
        else:                                                                           # maenv:772-805
            dones = {1: True, -1: True, "__all__": True}
            ff = hv.get('final_fobs')
            obs = {1: self._obs_dict(hv['final_obs'][0, 0], ff[0, 0] if ff is not None else None, hv['mask'][0], 1),
                   -1: self._obs_dict(hv['final_obs'][0, 1], ff[0, 1] if ff is not None else None, hv['mask'][0], -1)}
            infos = {1: {}, -1: {}}
            rew = hv['reward'][0]
            if flags[3]:
                rewards = {1: 0, -1: 0}
                for p in (1, -1):
                    infos[p]['game_result_was_invalid'] = True
                    infos[p]['game_result'] = 'tied'
            else:
                r1, r2 = np.float32(rew[0]), np.float32(rew[1])
                for p in (1, -1):
                    infos[p]['game_result_was_invalid'] = False
                if r1 == 1:
                    infos[1]['game_result'], infos[-1]['game_result'] = 'won', 'lost'
                elif r1 == -1:
                    infos[1]['game_result'], infos[-1]['game_result'] = 'lost', 'won'
                else:
                    infos[1]['game_result'], infos[-1]['game_result'] = 'tied', 'tied'
                rewards = {1: r1, -1: r2}
            if self.penalize_ties and infos[1]['game_result'] == 'tied':               # maenv:803-805
                rewards = {1: -0.5, -1: -0.5}
        if self.random_player_assignment:                                              # maenv:819-823
            obs = {self.player_map(k): val for k, val in obs.items()}
            rewards = {self.player_map(k): val for k, val in rewards.items()}
            dones = {self.player_map(k): val for k, val in dones.items()}
            infos = {self.player_map(k): val for k, val in infos.items()}
        return obs, rewards, dones, infos

    @staticmethod
    def sample_random_valid_action(valid_actions_mask):                                # maenv:830-834
        flat = np.reshape(valid_actions_mask, -1)
        p = flat / np.sum(valid_actions_mask)
        return np.random.choice(range(len(flat)), p=p)

    def close(self):
        if getattr(self, '_hslab_host', None) and getattr(self._vec, '_h', None):
            self._vec._L.sgx_host_free(self._vec._h, self._hslab_host)
            self._hslab_host = None
        self._vec.close()
        self.base_env.close()


def make_stratego_env(env_config):                                                     # maenv:837-838
    return StrategoMultiAgentEnv(env_config)
