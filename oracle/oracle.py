"""ctypes binding of oracle/stratego_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of stratego_oracle.c).  The product package never does.

`OracleEnv` restates StrategoMultiAgentEnv (reference maenv:316-834) for
observation_mode=PARTIALLY_OBSERVABLE / extended channels on top of the C functions, with the same
dict outputs, so parity tests read like a drop-in comparison.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libstratego_oracle.so")
_lib = None

I64 = C.c_int64
P_I64 = C.POINTER(C.c_int64)
P_F32 = C.POINTER(C.c_float)
P_U8 = C.POINTER(C.c_uint8)
P_U64 = C.POINTER(C.c_uint64)


class StepResult(C.Structure):
    _fields_ = [("error", C.c_int32), ("done", C.c_int32), ("next_player", C.c_int32),
                ("ending_invalid", C.c_int32), ("reward_p1", C.c_float), ("reward_m1", C.c_float)]


class CVariant(C.Structure):
    _fields_ = [("rows", I64), ("cols", I64), ("max_turns", I64), ("usable_rows", I64),
                ("piece_amounts", I64 * 13), ("obstacles", P_U8), ("setups", P_U8), ("n_setups", I64)]


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "stratego_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        # STRATEGO_ORACLE_LIB selects another build of the same source (the `make asan` library for sanitizer runs)
        L = C.CDLL(os.environ.get('STRATEGO_ORACLE_LIB') or _LIB_PATH)
        L.so_action_size.restype = I64
        L.so_action_size.argtypes = [I64, I64]
        L.so_spatial_channels.restype = I64
        L.so_spatial_channels.argtypes = [I64, I64]
        L.so_action_1d_from_positions.restype = I64
        L.so_action_1d_from_positions.argtypes = [I64] * 6
        L.so_action_spatial_from_positions.restype = C.c_int
        L.so_action_spatial_from_positions.argtypes = [I64] * 6 + [P_I64]
        L.so_action_positions_from_spatial.restype = None
        L.so_action_positions_from_spatial.argtypes = [I64] * 5 + [P_I64]
        L.so_action_1d_from_spatial.restype = I64
        L.so_action_1d_from_spatial.argtypes = [I64] * 5
        L.so_action_positions_from_1d.restype = C.c_int
        L.so_action_positions_from_1d.argtypes = [I64] * 3 + [P_I64]
        L.so_action_spatial_from_1d.restype = C.c_int
        L.so_action_spatial_from_1d.argtypes = [I64] * 3 + [P_I64]
        L.so_create_initial_state.restype = None
        L.so_create_initial_state.argtypes = [I64, I64, P_I64, P_I64, P_I64, I64, P_I64]
        for f in (L.so_valid_moves_spatial, L.so_valid_moves_1d, L.so_state_from_player_perspective):
            f.restype = None
            f.argtypes = [I64, I64, P_I64, I64, P_I64]
        L.so_action_1d_from_player_perspective.restype = I64
        L.so_action_1d_from_player_perspective.argtypes = [I64] * 4
        L.so_is_move_valid_by_position.restype = C.c_int
        L.so_is_move_valid_by_position.argtypes = [I64, I64, P_I64, I64, I64, I64, I64, I64, C.c_int]
        L.so_is_move_valid_by_1d.restype = C.c_int
        L.so_is_move_valid_by_1d.argtypes = [I64, I64, P_I64, I64, I64, C.c_int]
        L.so_game_ended.restype = C.c_float
        L.so_game_ended.argtypes = [I64, I64, P_I64, I64]
        L.so_game_result_is_invalid.restype = C.c_int
        L.so_game_result_is_invalid.argtypes = [I64, I64, P_I64]
        L.so_next_state.restype = C.c_int
        L.so_next_state.argtypes = [I64, I64, P_I64, I64, I64, C.c_int, P_I64]
        for f in (L.so_po_obs_extended, L.so_fo_obs_extended, L.so_po_obs_original, L.so_fo_obs_original):
            f.restype = None
            f.argtypes = [I64, I64, P_I64, I64, P_F32]
        for f in (L.so_p_obs_norm_constants, L.so_f_obs_norm_constants, L.so_p_obs_norm_constants_original,
                  L.so_f_obs_norm_constants_original):
            f.restype = None
            f.argtypes = [P_I64, P_F32, P_F32]
        L.so_env_current_obs3.restype = None
        L.so_env_current_obs3.argtypes = [I64, I64, P_I64, I64, C.c_int, P_F32, P_F32, P_F32, P_F32, P_U8, P_F32, P_F32]
        L.so_env_step3.restype = None
        L.so_env_step3.argtypes = [I64, I64, P_I64, P_I64, I64, C.c_int, C.c_int, P_F32, P_F32, P_F32, P_F32, P_U8, P_F32, P_F32,
                                   C.POINTER(StepResult)]
        L.so_normalize_obs.restype = None
        L.so_normalize_obs.argtypes = [I64, I64, P_F32, P_F32, P_F32]
        L.so_env_current_obs.restype = None
        L.so_env_current_obs.argtypes = [I64, I64, P_I64, I64, P_F32, P_F32, P_U8, P_F32]
        L.so_env_current_obs2.restype = None
        L.so_env_current_obs2.argtypes = [I64, I64, P_I64, I64, P_F32, P_F32, P_F32, P_F32, P_U8, P_F32, P_F32]
        L.so_env_step2.restype = None
        L.so_env_step2.argtypes = [I64, I64, P_I64, P_I64, I64, C.c_int, P_F32, P_F32, P_F32, P_F32, P_U8, P_F32, P_F32,
                                   C.POINTER(StepResult)]
        L.so_env_step.restype = None
        L.so_env_step.argtypes = [I64, I64, P_I64, P_I64, I64, C.c_int, P_F32, P_F32, P_U8, P_F32, C.POINTER(StepResult)]
        L.so_rng.restype = C.c_uint64
        L.so_rng.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.so_rng_below.restype = C.c_uint32
        L.so_rng_below.argtypes = [C.c_uint64, C.c_uint32]
        L.so_sample_setup.restype = None
        L.so_sample_setup.argtypes = [C.POINTER(CVariant), C.c_uint64, C.c_uint64, C.c_uint64, P_I64, P_I64]
        L.so_reset_env.restype = None
        L.so_reset_env.argtypes = [C.POINTER(CVariant), C.c_uint64, C.c_uint64, C.c_uint64, P_I64]
        L.so_sample_action.restype = I64
        L.so_sample_action.argtypes = [P_U8, I64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.so_fnv1a.restype = C.c_uint64
        L.so_fnv1a.argtypes = [C.c_uint64, C.c_void_p, I64]
        L.so_rollout.restype = I64
        L.so_rollout.argtypes = [C.POINTER(CVariant), C.c_uint64, I64, I64, I64, C.c_int, P_U64, P_I64]
        L.so_rollout_ex.restype = I64
        L.so_rollout_ex.argtypes = [C.POINTER(CVariant), C.c_uint64, I64, I64, I64, I64, C.c_int, C.c_int, P_U64, P_I64, P_U64, P_I64,
                                    C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class OracleRules:
    """Pure functions of (state int64[34,R,C], player, action): mirrors StrategoProceduralEnv (penv:20-181)."""

    def __init__(self, rows, columns):
        if rows < 3 or columns < 3:
            raise ValueError("Both rows and columns have to be at least 3")
        self.rows, self.columns = int(rows), int(columns)
        L = lib()
        self.action_size = int(L.so_action_size(rows, columns))
        self.K = int(L.so_spatial_channels(rows, columns))
        self.spatial_action_size = (self.rows, self.columns, self.K)

    def create_initial_state(self, obstacle_map, player_1_initial_piece_map, player_2_initial_piece_map, max_turns):
        R, Cc = self.rows, self.columns
        ob, m1, m2 = _i64(obstacle_map), _i64(player_1_initial_piece_map), _i64(player_2_initial_piece_map)
        for m in (ob, m1, m2):
            if m.shape != (R, Cc):
                raise ValueError("map needs to be of shape {}".format((R, Cc)))
        st = np.zeros((34, R, Cc), dtype=np.int64)
        lib().so_create_initial_state(R, Cc, _p(ob, P_I64), _p(m1, P_I64), _p(m2, P_I64), int(max_turns), _p(st, P_I64))
        return st

    def get_action_1d_index_from_positions(self, sr, sc, er, ec):
        return int(lib().so_action_1d_from_positions(self.rows, self.columns, sr, sc, er, ec))

    def get_action_positions_from_1d_index(self, idx):
        out = np.zeros(4, dtype=np.int64)
        if lib().so_action_positions_from_1d(self.rows, self.columns, int(idx), _p(out, P_I64)):
            raise ValueError("Action is a no-op so it doesn't translate to an actual action")
        return tuple(int(x) for x in out)

    def get_action_spatial_index_from_positions(self, sr, sc, er, ec):
        out = np.zeros(3, dtype=np.int64)
        if lib().so_action_spatial_from_positions(self.rows, self.columns, sr, sc, er, ec, _p(out, P_I64)):
            raise ValueError("diagonal or null move")
        return tuple(int(x) for x in out)

    def get_action_positions_from_spatial_index(self, spatial_index):
        out = np.zeros(4, dtype=np.int64)
        r, c, ch = (int(x) for x in spatial_index)
        lib().so_action_positions_from_spatial(self.rows, self.columns, r, c, ch, _p(out, P_I64))
        return tuple(int(x) for x in out)

    def get_action_1d_index_from_spatial_index(self, spatial_index):
        r, c, ch = (int(x) for x in spatial_index)
        return int(lib().so_action_1d_from_spatial(self.rows, self.columns, r, c, ch))

    def get_action_spatial_index_from_1d_index(self, idx):
        out = np.zeros(3, dtype=np.int64)
        if lib().so_action_spatial_from_1d(self.rows, self.columns, int(idx), _p(out, P_I64)):
            raise ValueError("no spatial index")
        return tuple(int(x) for x in out)

    def get_action_1d_index_from_player_perspective(self, action_index, player):
        return int(lib().so_action_1d_from_player_perspective(self.rows, self.columns, int(action_index), int(player)))

    def get_valid_moves_as_spatial_mask(self, state, player):
        st = _i64(state)
        out = np.zeros(self.spatial_action_size, dtype=np.int64)
        lib().so_valid_moves_spatial(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_I64))
        return out

    def get_valid_moves_as_1d_mask(self, state, player):
        st = _i64(state)
        out = np.zeros(self.action_size, dtype=np.int64)
        lib().so_valid_moves_1d(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_I64))
        return out

    def get_state_from_player_perspective(self, state, player):
        st = _i64(state)
        out = np.zeros_like(st)
        lib().so_state_from_player_perspective(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_I64))
        return out

    def is_move_valid_by_position(self, state, player, sr, sc, er, ec, allow_piece_oscillation=False):
        st = _i64(state)
        return bool(lib().so_is_move_valid_by_position(self.rows, self.columns, _p(st, P_I64), int(player), int(sr),
                                                       int(sc), int(er), int(ec), int(allow_piece_oscillation)))

    def is_move_valid_by_1d_index(self, state, player, action_index, allow_piece_oscillation=False):
        st = _i64(state)
        return bool(lib().so_is_move_valid_by_1d(self.rows, self.columns, _p(st, P_I64), int(player), int(action_index),
                                                 int(allow_piece_oscillation)))

    def get_next_state(self, state, player, action_index, allow_piece_oscillation=False):
        st = _i64(state)
        out = np.zeros_like(st)
        if lib().so_next_state(self.rows, self.columns, _p(st, P_I64), int(player), int(action_index),
                               int(allow_piece_oscillation), _p(out, P_I64)):
            raise ValueError("Couldn't get the next state because the move wasn't valid.")
        return out, -player

    def get_game_ended(self, state, player):
        st = _i64(state)
        return float(lib().so_game_ended(self.rows, self.columns, _p(st, P_I64), int(player)))

    def get_game_result_is_invalid(self, state):
        st = _i64(state)
        return bool(lib().so_game_result_is_invalid(self.rows, self.columns, _p(st, P_I64)))

    def get_partially_observable_observation_extended_channels(self, state, player):
        st = _i64(state)
        out = np.zeros((self.rows, self.columns, 67), dtype=np.float32)
        lib().so_po_obs_extended(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_F32))
        return out

    def get_fully_observable_observation_extended_channels(self, state, player):
        st = _i64(state)
        out = np.zeros((self.rows, self.columns, 79), dtype=np.float32)
        lib().so_fo_obs_extended(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_F32))
        return out


    def get_partially_observable_observation(self, state, player):
        """Deprecated 32-layer observation (impl:1153-1197)."""
        st = _i64(state)
        out = np.zeros((self.rows, self.columns, 32), dtype=np.float32)
        lib().so_po_obs_original(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_F32))
        return out

    def get_fully_observable_observation(self, state, player):
        """Deprecated 33-layer observation (impl:1075-1123)."""
        st = _i64(state)
        out = np.zeros((self.rows, self.columns, 33), dtype=np.float32)
        lib().so_fo_obs_original(self.rows, self.columns, _p(st, P_I64), int(player), _p(out, P_F32))
        return out


def piece_amounts_array(piece_counts):
    """(count of code 1, ..., count of code 12) -> int64[13] indexed by piece code."""
    a = np.zeros(13, dtype=np.int64)
    a[1:] = np.asarray(piece_counts, dtype=np.int64)
    return a


def f_obs_norm_constants(piece_counts, original=False):
    n = 33 if original else 79
    mids = np.zeros(n, dtype=np.float32)
    ranges = np.zeros(n, dtype=np.float32)
    pa = piece_amounts_array(piece_counts)
    fn = lib().so_f_obs_norm_constants_original if original else lib().so_f_obs_norm_constants
    fn(_p(pa, P_I64), _p(mids, P_F32), _p(ranges, P_F32))
    return mids, ranges


def p_obs_norm_constants(piece_counts, original=False):
    n = 32 if original else 67
    mids = np.zeros(n, dtype=np.float32)
    ranges = np.zeros(n, dtype=np.float32)
    pa = piece_amounts_array(piece_counts)
    fn = lib().so_p_obs_norm_constants_original if original else lib().so_p_obs_norm_constants
    fn(_p(pa, P_I64), _p(mids, P_F32), _p(ranges, P_F32))
    return mids, ranges


class OracleEnv:
    """N=1 restatement of StrategoMultiAgentEnv for PARTIALLY_OBSERVABLE mode (maenv:316-834).

    Constructed from plain numbers (no product import): rows, columns, max_turns, obstacle cells,
    piece counts (codes 1..12).  `reset(p1_map, p2_map)` takes own-side piece maps like
    StrategoProceduralEnv.create_initial_state, or `reset(initial_state_override=state)`.
    """

    MASK = 'valid_actions_mask'
    POBS = 'partial_observation'
    FOBS = 'full_observation'

    def __init__(self, rows, columns, max_turns, obstacle_locations, piece_counts, penalize_ties=False,
                 observation_mode='partially_observable', obs_channel_mode='extended'):
        assert observation_mode in ('partially_observable', 'fully_observable', 'both_observations')
        self.mode = observation_mode
        self.original = obs_channel_mode != 'extended'          # maenv:368
        self.p_ch, self.f_ch = (32, 33) if self.original else (67, 79)
        self.rules = OracleRules(rows, columns)
        self.rows, self.columns, self.max_turns = int(rows), int(columns), int(max_turns)
        self.K = self.rules.K
        self.obstacles = np.zeros((rows, columns), dtype=np.int64)
        for r, c in obstacle_locations:
            self.obstacles[r, c] = 1
        self.piece_counts = tuple(int(x) for x in piece_counts)
        self.penalize_ties = bool(penalize_ties)
        self.mids, self.ranges = p_obs_norm_constants(self.piece_counts, self.original)
        self.f_mids, self.f_ranges = f_obs_norm_constants(self.piece_counts, self.original)
        self.state = None
        self.player = 1

    def _obs(self, player):
        R, Cc, K = self.rows, self.columns, self.K
        mask = np.zeros((R, Cc, K), dtype=np.uint8)
        pobs = np.zeros((R, Cc, self.p_ch), dtype=np.float32)
        fobs = np.zeros((R, Cc, self.f_ch), dtype=np.float32)
        lib().so_env_current_obs3(R, Cc, _p(self.state, P_I64), int(player), int(self.original), _p(self.mids, P_F32),
                                  _p(self.ranges, P_F32), _p(self.f_mids, P_F32), _p(self.f_ranges, P_F32), _p(mask, P_U8), _p(pobs, P_F32), _p(fobs, P_F32))
        return self._pack(mask, pobs, fobs)

    def _pack(self, mask, pobs, fobs):
        d = {self.MASK: mask.astype(np.int64)}
        if self.mode != 'fully_observable':
            d[self.POBS] = pobs
        if self.mode != 'partially_observable':
            d[self.FOBS] = fobs
        return d

    def reset(self, p1_map=None, p2_map=None, initial_state_override=None, first_player_override=None):
        if initial_state_override is not None:
            self.state = _i64(initial_state_override).copy()
        else:
            self.state = self.rules.create_initial_state(self.obstacles, p1_map, p2_map, self.max_turns)
        self.player = 1 if first_player_override is None else int(first_player_override)
        return {self.player: self._obs(self.player)}

    def step(self, action_dict):
        assert self.player in action_dict and -self.player not in action_dict  # maenv:678-679
        action = int(action_dict[self.player])
        R, Cc, K = self.rows, self.columns, self.K
        mask = np.zeros((2, R, Cc, K), dtype=np.uint8)
        pobs = np.zeros((2, R, Cc, self.p_ch), dtype=np.float32)
        fobs = np.zeros((2, R, Cc, self.f_ch), dtype=np.float32)
        res = StepResult()
        pl = C.c_int64(self.player)
        lib().so_env_step3(R, Cc, _p(self.state, P_I64), C.byref(pl), action, int(self.penalize_ties), int(self.original),
                           _p(self.mids, P_F32), _p(self.ranges, P_F32), _p(self.f_mids, P_F32), _p(self.f_ranges, P_F32),
                           _p(mask, P_U8), _p(pobs, P_F32), _p(fobs, P_F32), C.byref(res))
        if res.error:
            raise ValueError("Couldn't get the next state because the move wasn't valid.")
        self.player = int(pl.value)
        if not res.done:
            obs = {self.player: self._pack(mask[0], pobs[0], fobs[0])}
            return obs, {self.player: 0}, {self.player: False, "__all__": False}, {}
        obs = {1: self._pack(mask[0], pobs[0], fobs[0]), -1: self._pack(mask[1], pobs[1], fobs[1])}
        rewards = {1: float(res.reward_p1), -1: float(res.reward_m1)}
        dones = {1: True, -1: True, "__all__": True}
        inv = bool(res.ending_invalid)
        if inv:
            r1 = 'tied'
        else:
            r1 = 'won' if res.reward_p1 == 1 else ('lost' if res.reward_p1 == -1 else 'tied')
        r2 = {'won': 'lost', 'lost': 'won', 'tied': 'tied'}[r1]
        if self.penalize_ties and r1 == 'tied':
            rewards = {1: -0.5, -1: -0.5}
        infos = {1: {'game_result_was_invalid': inv, 'game_result': r1},
                 -1: {'game_result_was_invalid': inv, 'game_result': r2}}
        return obs, rewards, dones, infos


def make_cvariant(rows, columns, max_turns, obstacle_locations, piece_counts, usable_rows, setups=None):
    """Build the C struct for the rollout harness; keeps numpy buffers alive on the returned object."""
    v = CVariant()
    v.rows, v.cols, v.max_turns, v.usable_rows = rows, columns, max_turns, usable_rows
    pa = piece_amounts_array(piece_counts)
    for i in range(13):
        v.piece_amounts[i] = int(pa[i])
    ob = np.zeros((rows, columns), dtype=np.uint8)
    for r, c in obstacle_locations:
        ob[r, c] = 1
    v._ob = ob
    v.obstacles = _p(ob, P_U8)
    if setups is not None:
        s = np.ascontiguousarray(setups, dtype=np.uint8)
        assert s.ndim == 2 and s.shape[1] == usable_rows * columns
        v._setups = s
        v.setups = _p(s, P_U8)
        v.n_setups = s.shape[0]
    else:
        v.setups = None
        v.n_setups = 0
    return v


def rng(seed, g, j, stream, t):
    return int(lib().so_rng(seed, g, j, stream, t))


def rng_below(r, n):
    return int(lib().so_rng_below(r, n))


def sample_setup(cv, seed, g, j):
    R, Cc = int(cv.rows), int(cv.cols)
    m1 = np.zeros((R, Cc), dtype=np.int64)
    m2 = np.zeros((R, Cc), dtype=np.int64)
    lib().so_sample_setup(C.byref(cv), seed, g, j, _p(m1, P_I64), _p(m2, P_I64))
    return m1, m2


def reset_state(cv, seed, g, j):
    st = np.zeros((34, int(cv.rows), int(cv.cols)), dtype=np.int64)
    lib().so_reset_env(C.byref(cv), seed, g, j, _p(st, P_I64))
    return st


def sample_action(mask_u8, seed, g, j, turn):
    m = np.ascontiguousarray(mask_u8, dtype=np.uint8).reshape(-1)
    return int(lib().so_sample_action(_p(m, P_U8), m.size, seed, g, j, turn))


def rollout(cv, seed, g0, n_envs, n_steps, threads=1):
    """Returns (total_steps, digests uint64[n_envs], games_finished int64[n_envs])."""
    dig = np.zeros(n_envs, dtype=np.uint64)
    fin = np.zeros(n_envs, dtype=np.int64)
    total = lib().so_rollout(C.byref(cv), seed, g0, n_envs, n_steps, threads, _p(dig, P_U64), _p(fin, P_I64))
    return int(total), dig, fin


def rollout_ex(cv, seed, g0, n_envs, n_steps, skip=0, both=False, threads=1, want_states=False):
    """so_rollout_ex: rolling digests over steps skip.., digests of the last step alone, games finished, final info
    int32 [n,4] = (turn, game number, game_over, player) and (optionally) the final int64 states."""
    dig = np.zeros(n_envs, dtype=np.uint64)
    last = np.zeros(n_envs, dtype=np.uint64)
    fin = np.zeros(n_envs, dtype=np.int64)
    info = np.zeros((n_envs, 4), dtype=np.int32)
    states = np.zeros((n_envs, 34, int(cv.rows), int(cv.cols)), dtype=np.int64) if want_states else None
    lib().so_rollout_ex(C.byref(cv), seed, g0, n_envs, n_steps, skip, 1 if both else 0, threads, _p(dig, P_U64), _p(fin, P_I64),
                        _p(last, P_U64), _p(states, P_I64) if want_states else None, info.ctypes.data_as(C.POINTER(C.c_int32)))
    return {'digests': dig, 'last_digests': last, 'games_finished': fin, 'info': info, 'states': states}


def step_digest(mask, obs, reward, done, player, ending_invalid, fobs=None, h=None):
    """One step's contribution to so_rollout's digest (continuing from h, default: the FNV offset basis)."""
    tail = np.asarray([int(done), int(player), int(ending_invalid), 0], dtype=np.int32)
    data = np.ascontiguousarray(mask, dtype=np.uint8).tobytes() + np.ascontiguousarray(obs, dtype=np.float32).tobytes()
    if fobs is not None:
        data += np.ascontiguousarray(fobs, dtype=np.float32).tobytes()
    data += np.ascontiguousarray(reward, dtype=np.float32).tobytes() + tail.tobytes()
    return fnv1a(FNV_OFFSET if h is None else h, data)


FNV_OFFSET = 0xCBF29CE484222325


def fnv1a(h, data: bytes):
    """FNV-1a 64 continuation over `data` (the digest so_rollout() keeps per env)."""
    buf = C.create_string_buffer(data, len(data))
    return int(lib().so_fnv1a(h, C.cast(buf, C.c_void_p), len(data)))
