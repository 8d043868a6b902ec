"""BatchedStrategoProceduralEnv: the reference's functional operator API (StrategoProceduralEnv,
game/stratego_procedural_env.py:20-181) for a batch of caller-provided states.

Every method is a pure function of (states int64 [N,34,R,C] in the reference layout, players [N], actions [N]) and returns
fresh tensors, like the reference's methods do for one state.  Game logic runs in the HIP kernels: the states are imported
into a scratch handle (`sgx_import_state`), stepped / observed there, and exported again; tree-search style callers keep
their states on the device.  Where the reference raises ValueError for an invalid move, `get_next_state` returns a
`valid` mask and leaves that state unchanged.

Limits (documented in DESIGN.md): states must be reachable ones -- at most two non-zero recent-move cells per player and
no more captured pieces than pieces exist; the obstacle layer must equal the variant's (it is a per-handle constant).  Every
import reports which states it had to alter (`last_sanitised`, uint8 [N]; `strict=True` raises ValueError instead).

Every call imports the states it is given.  A caller that asks several questions about the SAME batch can hold it loaded:

    with env.loaded(states, players):
        mask = env.get_valid_moves_as_1d_mask(states, players)
        obs = env.get_partially_observable_observation_extended_channels(states, players)

(inside the scope, calls given the same tensor OBJECTS skip the import; the caller promises not to write to them meanwhile).
Search callers avoid the 27 KB int64 layout altogether with PackedStates (pack / expand / unpack below).
"""
import contextlib

import numpy as np
import torch

from . import _lib, index_algebra as ia
from .config import NUM_STATE_LAYERS, get_variant
from .vec_env import VecStrategoEnv

# layer pairs swapped by the perspective flip (impl:645-675)
_SWAP_A = [0, 3, 6, 32] + list(range(8, 20))
_SWAP_B = [1, 4, 7, 33] + list(range(20, 32))


def _ptr_or_none(t):
    return t.data_ptr() if t is not None else None


class BatchedStrategoProceduralEnv:
    def __init__(self, version, batch_size, device=0):
        self.variant = get_variant(version)
        v = self.variant
        if v.rows < 3 or v.columns < 3:
            raise ValueError("Both rows and columns have to be at least 3")                        # penv:28-30
        self.rows, self.columns = v.rows, v.columns
        self.batch_size = int(batch_size)
        self.action_size = v.action_size                                                            # penv:34
        self.spatial_action_size = v.spatial_action_size                                            # penv:35
        self._vec = VecStrategoEnv(v, batch_size, device=device, human_inits=False)
        self.device = self._vec.device
        self._held = None            # (states, players) objects held loaded by a `with env.loaded(...)` scope
        self._scratch_is_held = False
        self._held_exact = True      # False: the held states include ones the packed record cannot carry -> no reuse inside the scope
        self.strict = False          # True: ValueError when an import had to alter a state (sanitised)
        self.last_sanitised = torch.zeros((self.batch_size,), dtype=torch.uint8, device=self.device)
        self._obstacles = torch.from_numpy(v.obstacle_map().astype(np.int64)).to(self.device)

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _players(self, players):
        p = torch.as_tensor(players).to(device=self.device, dtype=torch.int8).reshape(self.batch_size).contiguous()
        return p

    def _load(self, states, players):
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64).contiguous()
        if tuple(st.shape) != (self.batch_size, NUM_STATE_LAYERS, self.rows, self.columns):
            raise ValueError("states must have shape (batch, 34, rows, columns)")
        pl = self._players(players)
        vec = self._vec
        if self._held is not None and self._scratch_is_held and states is self._held[0] and players is self._held[1]:
            return st, pl                                   # inside `with env.loaded(states, players)`: already in the scratch handle
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_import_state_checked(vec._h, st.data_ptr(), pl.data_ptr(), self.last_sanitised.data_ptr(),
                                                       vec._stream()), vec._L)
        self._scratch_is_held = self._held is not None and self._held_exact and states is self._held[0] and players is self._held[1]
        if self.strict and bool(self.last_sanitised.any()):
            raise ValueError("state is not one the packed record can carry (unreachable by play): see sgx_import_state_checked")
        return st, pl

    def _scratch_changed(self):
        """The scratch handle no longer holds the states of an enclosing `loaded` scope (a move was applied to it)."""
        self._scratch_is_held = False

    @contextlib.contextmanager
    def loaded(self, states, players):
        """Opt-in: import `states` once and let the calls inside the scope that are given the same tensor objects reuse the
        import.  The caller promises not to modify the tensors inside the scope (writes through other libraries or streams
        are invisible to this class); outside a scope every call imports.

        Results do not depend on the scope: if the import had to alter any of the states -- general (unreachable) states the packed
        record cannot carry -- nothing is reused and the calls inside the scope import like calls outside it, through sgx_step_states'
        general-state pass.  `last_sanitised` therefore means the same on both paths: what still had to be altered after every pass
        the call ran; `strict=True` raises on exactly that."""
        prev = (self._held, self._held_exact)
        self._held = (states, players)
        self._scratch_is_held = False
        self._held_exact = True
        try:
            strict, self.strict = self.strict, False
            try:
                self._load(states, players)
            finally:
                self.strict = strict
            if bool(self.last_sanitised.any()):              # one device -> host round trip per scope
                self._held_exact = False
                self._scratch_is_held = False
            yield self
        finally:
            self._held, self._held_exact = prev
            self._scratch_is_held = False

    def _in_loaded_scope(self, states, players):
        return self._held is not None and self._scratch_is_held and states is self._held[0] and players is self._held[1]

    def _step_states(self, states, players, actions, flags, export=False, mask_out=None, positions=False, out=None, obs_out=None, fobs_out=None):
        """sgx_step_states: import -> step (actions given) or observe (actions None) -> optional export, one library call (one
        launch on boards of more than 32 cells).  -> (new_states, new_players) or None."""
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64).contiguous()
        if tuple(st.shape) != (self.batch_size, NUM_STATE_LAYERS, self.rows, self.columns):
            raise ValueError("states must have shape (batch, 34, rows, columns)")
        pl = self._players(players)
        vec = self._vec
        if out is not None:                                  # caller-provided successor tensors (benchmarks: same memory every call)
            new_states, new_players = out
        else:
            new_states = torch.empty_like(st) if export else None
            new_players = torch.empty((self.batch_size,), dtype=torch.int8, device=self.device) if export else None
        io = vec._fill_io(actions if actions is not None else vec.next_actions, False, False, False, flags)
        io.auto_reset = 0
        if actions is None:
            io.actions_dev = None
        io.mask_dev = mask_out.data_ptr() if mask_out is not None else None
        io.obs_dev = obs_out.data_ptr() if obs_out is not None else None
        io.fobs_dev = fobs_out.data_ptr() if fobs_out is not None else None
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_step_states(vec._h, st.data_ptr(), pl.data_ptr(), self.last_sanitised.data_ptr(), io,
                                              _ptr_or_none(new_states), _ptr_or_none(new_players), 2, vec._stream()), vec._L)
        vec._next_actions_fresh = False
        self._scratch_is_held = (actions is None and self._held is not None and self._held_exact and states is self._held[0] and players is self._held[1])
        if self.strict and bool(self.last_sanitised.any()):
            raise ValueError("state is not one the packed record can carry (unreachable by play): see sgx_import_state_checked")
        return (new_states, new_players) if export else None

    def _mask_in_state_coordinates(self, one_dim):
        """Mask of the loaded states' movers, indexed in the states' own coordinates (no perspective flip), rendered by the
        kernel: SGX_STEP_MASK_1D -> uint8 [N, action_size], SGX_STEP_MASK_STATE_COORDS -> uint8 [N, R, C, K]."""
        vec = self._vec
        shape = (self.batch_size, self.action_size) if one_dim else (self.batch_size,) + tuple(self.spatial_action_size)
        out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_observe(vec._h, None, None, out.data_ptr(), None,
                                          _lib.STEP_MASK_1D if one_dim else _lib.STEP_MASK_STATE_COORDS, vec._stream()), vec._L)
        return out

    # ---- state construction / transition -----------------------------------------------------------------------
    def create_initial_state(self, player_1_initial_piece_maps, player_2_initial_piece_maps):      # penv:38-60
        """own-side piece maps int [N,R,C] -> states (obstacle map and max_turns come from the variant)."""
        self._vec.reset(player_1_initial_piece_maps, player_2_initial_piece_maps)
        self._scratch_changed()
        st, _ = self._vec.export_state()
        return st

    def get_next_state(self, states, players, action_indices, allow_piece_oscillation=False):      # penv:148-155
        """-> (new_states, new_players int8, valid bool).  action_indices: absolute 1-D indices (impl:262-277)."""
        flags = _lib.STEP_ACTIONS_1D | (_lib.STEP_ALLOW_OSCILLATION if allow_piece_oscillation else 0)
        a = torch.as_tensor(action_indices).to(device=self.device, dtype=torch.int32).reshape(self.batch_size).contiguous()
        if self._held is not None and self._scratch_is_held and states is self._held[0] and players is self._held[1]:
            self._vec.step(a, emit_obs=False, emit_mask=False, flags=flags)          # inside `loaded`: the states are in the handle
            self._scratch_changed()
            new_states, new_players = self._vec.export_state()
            return new_states, new_players, self._vec.invalid_action == 0
        new_states, new_players = self._step_states(states, players, a, flags, export=True)
        return new_states, new_players, self._vec.invalid_action == 0

    def is_move_valid_by_1d_index(self, states, players, action_indices, allow_piece_oscillation=False):   # penv:94-99
        flags = _lib.STEP_ACTIONS_1D | (_lib.STEP_ALLOW_OSCILLATION if allow_piece_oscillation else 0)
        a = torch.as_tensor(action_indices).to(device=self.device, dtype=torch.int32).reshape(self.batch_size).contiguous()
        if self._in_loaded_scope(states, players):
            self._vec.step(a, emit_obs=False, emit_mask=False, flags=flags)
            self._scratch_changed()
        else:
            self._step_states(states, players, a, flags)
        return self._vec.invalid_action == 0

    def is_move_valid_by_position(self, states, players, start_r, start_c, end_r, end_c, allow_piece_oscillation=False):  # penv:87-92
        pos = torch.stack([torch.as_tensor(x).to(device=self.device, dtype=torch.int32).reshape(self.batch_size)
                           for x in (start_r, start_c, end_r, end_c)], dim=1).contiguous()
        flags = _lib.STEP_ACTIONS_POSITIONS | (_lib.STEP_ALLOW_OSCILLATION if allow_piece_oscillation else 0)
        if self._in_loaded_scope(states, players):
            self._vec.step(pos.view(-1), emit_obs=False, emit_mask=False, flags=flags)
            self._scratch_changed()
        else:
            self._step_states(states, players, pos.view(-1), flags)       # (one call: states the packed record cannot carry are redone exactly)
        return self._vec.invalid_action == 0

    # ---- masks ---------------------------------------------------------------------------------------------------
    def get_valid_moves_as_spatial_mask(self, states, players):                                     # penv:127-128
        """uint8 [N,R,C,K] in the coordinates of the given states (no perspective flip), like impl:399-517."""
        if not self._in_loaded_scope(states, players):
            out = torch.empty((self.batch_size,) + tuple(self.spatial_action_size), dtype=torch.uint8, device=self.device)
            self._step_states(states, players, None, _lib.STEP_MASK_STATE_COORDS, mask_out=out)
            return out
        return self._mask_in_state_coordinates(one_dim=False)

    def get_valid_moves_as_1d_mask(self, states, players, player_perspective=False):                # penv:74-80
        """uint8 [N, action_size] in the coordinates of the given states, like impl:520-642 (last element = no-op).
        player_perspective=True first flips the states of player -1 (penv:76-77) and then, like the reference, still
        asks for player -1's moves on the flipped state."""
        if player_perspective:
            states = self.get_state_from_player_perspective(states, players)
        if not self._in_loaded_scope(states, players):
            out = torch.empty((self.batch_size, self.action_size), dtype=torch.uint8, device=self.device)
            self._step_states(states, players, None, _lib.STEP_MASK_1D, mask_out=out)
            return out
        return self._mask_in_state_coordinates(one_dim=True)

    def get_dict_of_valid_moves_by_position(self, states, players):                                 # penv:82-85 / impl:1400-1429
        """One dict per state: "start_r,start_c" -> [[end_r, end_c], ...] in ascending 1-D index order."""
        masks = self.get_valid_moves_as_1d_mask(states, players).cpu().numpy()
        out = []
        for m in masks:
            d = {}
            for idx in np.flatnonzero(m):
                if idx == self.action_size - 1:                                                      # impl:355-367
                    raise ValueError("Action is a no-op so it doesn't translate to an actual action")
                sr, sc, er, ec = (int(x) for x in ia.positions_from_1d(self.rows, self.columns, int(idx)))
                d.setdefault("{},{}".format(sr, sc), []).append([er, ec])
            out.append(d)
        return out

    def get_serializable_string_for_fully_observable_state(self, states):                           # penv:175-177
        """pickle of the raw 33-channel observation from player 1's side, one bytes object per state."""
        from pickle import dumps
        obs = self.get_fully_observable_observation(states, np.ones(self.batch_size, dtype=np.int8)).cpu().numpy()
        return [dumps(np.ascontiguousarray(o)) for o in obs]

    def get_serializable_string_for_partially_observable_state(self, states):                       # penv:179-181
        from pickle import dumps
        obs = self.get_partially_observable_observation(states, np.ones(self.batch_size, dtype=np.int8)).cpu().numpy()
        return [dumps(np.ascontiguousarray(o)) for o in obs]

    @staticmethod
    def print_board_to_console(state, partially_observable=False, hide_still_piece_markers=True):  # penv:183-214
        """Text rendering of ONE state (int [34,R,C]): player 1's pieces as positive codes, player -1's (true or
        partially-observable layer) as negative codes, "R" for obstacles; row and column 0 at the bottom right."""
        st = state.cpu().numpy() if isinstance(state, torch.Tensor) else np.asarray(state)
        _, rows, columns = st.shape
        enemy_layer = 4 if partially_observable else 1
        rule = "\n       " + "-" * (rows * 6 + 1)
        lines = ["    COL" + "".join("  {}  ".format(str(c).rjust(2)) for c in range(columns - 1, -1, -1)) + rule]
        for r in range(rows - 1, -1, -1):
            cells = []
            for c in range(columns - 1, -1, -1):
                item = ""
                if st[0, r, c] != 0:
                    item = str(st[0, r, c])
                elif st[enemy_layer, r, c] != 0:
                    item = str(-1 * st[enemy_layer, r, c])
                elif st[2, r, c] != 0:
                    item = "R"
                if not hide_still_piece_markers:
                    item += "a" if st[32, r, c] == 1 else ""
                    item += "b" if st[33, r, c] == 1 else ""
                cells.append(item.rjust(4) + " |")
            lines.append("Row {} |".format(str(r).rjust(2)) + "".join(cells) + rule)
        print("\n".join(lines))

    # ---- observations (raw, as the reference's operator layer returns them; maenv normalises afterwards) -----------
    def get_partially_observable_observation_extended_channels(self, states, players):             # penv:171-173
        return self._observe_raw(states, players, full=False, original=False)

    def get_fully_observable_observation_extended_channels(self, states, players):                 # penv:166-169
        return self._observe_raw(states, players, full=True, original=False)

    def _observe_raw(self, states, players, full, original):
        """One raw (un-normalised) observation kind into a fresh tensor; nothing else is rendered."""
        vec = self._vec
        if original:
            ch = _lib.FO_OBS_CHANNELS_ORIGINAL if full else _lib.PO_OBS_CHANNELS_ORIGINAL
        else:
            ch = _lib.FO_OBS_CHANNELS if full else _lib.PO_OBS_CHANNELS
        out = torch.empty((self.batch_size, self.rows, self.columns, ch), dtype=torch.float32, device=self.device)
        flags = _lib.STEP_RAW_OBS | (_lib.STEP_ORIGINAL_CHANNELS if original else 0)
        if not self._in_loaded_scope(states, players):
            # through sgx_step_states (observe): states the packed record cannot carry are redone exactly (every kind, boards <= 256 cells)
            self._step_states(states, players, None, flags, obs_out=None if full else out, fobs_out=out if full else None)
            return out
        self._load(states, players)
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_observe(vec._h, None if full else out.data_ptr(), out.data_ptr() if full else None, None, None,
                                          flags, vec._stream()), vec._L)
        return out

    def get_partially_observable_observation(self, states, players):                               # penv:162-164
        """Deprecated 32-channel observation holding piece values (impl:1153-1197), raw."""
        return self._observe_raw(states, players, full=False, original=True)

    def get_fully_observable_observation(self, states, players):                                   # penv:157-160
        """Deprecated 33-channel observation (impl:1075-1123), raw."""
        return self._observe_raw(states, players, full=True, original=True)

    # ---- pure tensor functions (no kernel needed) -------------------------------------------------------------------
    def get_state_from_player_perspective(self, states, players):                                   # penv:101-103 / impl:645-675
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64)
        pl = self._players(players)
        flipped = st.clone()
        f = st.flip(dims=(2, 3))
        flipped[:, _SWAP_A] = f[:, _SWAP_B]
        flipped[:, _SWAP_B] = f[:, _SWAP_A]
        flipped[:, 2] = f[:, 2]
        return torch.where((pl < 0).view(-1, 1, 1, 1), flipped, st)

    def get_game_ended(self, states, players):                                                      # penv:141-143 / impl:834-842
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64)
        pl = self._players(players).to(torch.float32)
        over, winner = st[:, 5, 0, 1] != 0, st[:, 5, 0, 2].to(torch.float32)
        val = torch.where(winner == 0, torch.full_like(winner, 1e-4), winner * pl)
        return torch.where(over, val, torch.zeros_like(val))

    def get_heuristic_rewards_from_move(self, states, players, action_indices, reward_matrix):     # impl:852-891
        """reward_matrix[moved own piece type, piece type on the destination] per state; 0 for the no-op.  Like the
        reference, the move is assumed valid (absolute 1-D indices, impl:262-277)."""
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64)
        pl = self._players(players)
        a = torch.as_tensor(action_indices).to(device=self.device, dtype=torch.int64).reshape(self.batch_size)
        rm = torch.as_tensor(reward_matrix).to(device=self.device, dtype=torch.float32)
        R, C = self.rows, self.columns
        noop = a == self.action_size - 1
        a = torch.where(noop, torch.zeros_like(a), a)
        start, k = a // (R + C), a % (R + C)                                                        # impl:350-383
        sr, sc = start // C, start % C
        er = torch.where(k < R, k, sr)
        ec = torch.where(k < R, sc, k - R)
        n = torch.arange(self.batch_size, device=self.device)
        own_layer = (pl < 0).to(torch.int64)                                                        # impl:172-174
        moved = st[n, own_layer, sr, sc]
        dest = st[n, 1 - own_layer, er, ec]
        return torch.where(noop, torch.zeros((), dtype=torch.float32, device=self.device), rm[moved, dest])

    def get_game_result_is_invalid(self, states):                                                   # penv:145-146 / impl:845-849
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64)
        return (st[:, 5, 0, 1] != 0) & (st[:, 5, 1, 1] != 0)

    # ---- index converters (batched numpy/torch-friendly; impl:262-396, 678-720) ------------------------------------------
    def get_action_1d_index_from_positions(self, sr, sc, er, ec):
        return ia.action_1d_from_positions(self.rows, self.columns, sr, sc, er, ec)

    def get_action_positions_from_1d_index(self, action_index):
        idx = np.asarray(action_index, dtype=np.int64)
        if np.any(idx == self.action_size - 1):
            raise ValueError("Action is a no-op so it doesn't translate to an actual action")     # impl:355-367
        return ia.positions_from_1d(self.rows, self.columns, idx)

    def get_action_1d_index_from_spatial_index(self, spatial_index):
        r, c, ch = spatial_index
        return ia.action_1d_from_spatial(self.rows, self.columns, r, c, ch)

    def get_action_spatial_index_from_positions(self, sr, sc, er, ec):
        r, c, ch = ia.spatial_from_positions(self.rows, self.columns, sr, sc, er, ec)
        if np.any(ch < 0):
            raise ValueError("diagonal or null move")                                              # impl:288-306
        return r, c, ch

    def get_action_positions_from_spatial_index(self, spatial_index):
        r, c, ch = spatial_index
        return ia.positions_from_spatial(self.rows, self.columns, r, c, ch)

    def get_action_positions_from_player_perspective(self, player, sr, sc, er, ec):
        if player == 1:
            return sr, sc, er, ec
        return ia.flip_positions(self.rows, self.columns, *(np.asarray(x, dtype=np.int64) for x in (sr, sc, er, ec)))

    def get_action_1d_index_from_player_perspective(self, action_index, player):
        return ia.action_1d_from_player_perspective(self.rows, self.columns, action_index, player)

    def get_action_spatial_index_from_1d_index(self, action_index):                                 # penv:135-139
        sr, sc, er, ec = self.get_action_positions_from_1d_index(action_index)
        return self.get_action_spatial_index_from_positions(sr, sc, er, ec)

    # ---- packed states: search nodes kept in the library's records (no int64 import / export per call) ---------------
    def new_packed(self, n=None):
        """An empty pool of `n` packed states of this variant (default: batch_size)."""
        return PackedStates(self.variant, self.batch_size if n is None else n, self.device)

    def pack(self, states, players, out=None):
        """int64 [n,34,R,C] + players -> PackedStates (`out` or a new pool); `sanitised` uint8 [n] reports altered states."""
        st = torch.as_tensor(states).to(device=self.device, dtype=torch.int64).contiguous()
        out = out if out is not None else self.new_packed(st.shape[0])
        out._vec.import_state_checked(st, players, out.sanitised)
        if self.strict and bool(out.sanitised.any()):
            raise ValueError("state is not one the packed record can carry (unreachable by play)")
        return out

    def close(self):
        self._vec.close()


class PackedStates:
    """A pool of n game states in the library's packed records (0.5 KB each for Barrage against 27 KB in the reference's int64
    layout), for tree-search callers of get_next_state (penv:148-155): nodes are expanded pool-to-pool with `expand`, copied with
    `copy_from`, and only converted to the reference layout when somebody wants to look at them (`unpack`)."""

    def __init__(self, version, n, device=0):
        self._vec = VecStrategoEnv(version, n, device=device, human_inits=False, outputs=False)     # records only: no observation / mask tensors
        self.n = int(n)
        self.device = self._vec.device
        self.sanitised = torch.zeros((self.n,), dtype=torch.uint8, device=self.device)

    def unpack(self):
        """-> (states int64 [n,34,R,C], players int8 [n])."""
        return self._vec.export_state()

    def copy_from(self, src, src_index=None, dst_index=None, n=None):
        """records src[src_index[i]] -> self[dst_index[i]] (index tensors int32 on the device, None = identity)."""
        vec = self._vec
        if src is self and (src_index is not None or dst_index is not None):
            raise ValueError("an indexed copy inside one pool would race (records read and rewritten by one launch): copy into another pool")
        si = None if src_index is None else torch.as_tensor(src_index).to(device=self.device, dtype=torch.int32).contiguous()
        di = None if dst_index is None else torch.as_tensor(dst_index).to(device=self.device, dtype=torch.int32).contiguous()
        if n is None:
            n = si.numel() if si is not None else di.numel() if di is not None else min(self.n, src.n)
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_copy_envs(vec._h, None if di is None else di.data_ptr(), src._vec._h,
                                            None if si is None else si.data_ptr(), int(n), vec._stream()), vec._L)
        return self

    def expand(self, parents, action_indices, parent_index=None, allow_piece_oscillation=False, mask_1d_out=None):
        """get_next_state for every slot i of this pool: self[i] = next_state(parents[parent_index[i]], action_indices[i])
        (absolute 1-D actions, impl:262-277).  -> (valid bool [n], players int8 [n] of the successors); where the move is invalid
        the slot holds a copy of the parent.  mask_1d_out: optional uint8 [n, action_size] receiving the successors'
        get_valid_moves_as_1d_mask in the same launch."""
        vec = self._vec
        if parents is self and parent_index is not None:
            raise ValueError("in-place expansion through parent_index would race (a parent may be overwritten before it is read): "
                             "expand into another pool")
        a = torch.as_tensor(action_indices).to(device=self.device, dtype=torch.int32).reshape(self.n).contiguous()
        pi = None if parent_index is None else torch.as_tensor(parent_index).to(device=self.device, dtype=torch.int32).reshape(self.n).contiguous()
        flags = _lib.STEP_ACTIONS_1D | (_lib.STEP_ALLOW_OSCILLATION if allow_piece_oscillation else 0)
        if mask_1d_out is not None:
            assert mask_1d_out.dtype == torch.uint8 and mask_1d_out.is_contiguous() and mask_1d_out.shape[0] == self.n
            flags |= _lib.STEP_MASK_1D
        io = vec._fill_io(a, False, False, False, flags)
        io.auto_reset = 0
        io.mask_dev = mask_1d_out.data_ptr() if mask_1d_out is not None else None
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_expand(vec._h, parents._vec._h, None if pi is None else pi.data_ptr(), io, vec._stream()), vec._L)
        vec._next_actions_fresh = False
        return vec.invalid_action == 0, vec.player

    def valid_moves_as_1d_mask(self):
        out = torch.empty((self.n, self._vec.variant.action_size), dtype=torch.uint8, device=self.device)
        vec = self._vec
        with torch.cuda.device(self.device):
            _lib.check(vec._L.sgx_observe(vec._h, None, None, out.data_ptr(), None, _lib.STEP_MASK_1D, vec._stream()), vec._L)
        return out

    def close(self):
        self._vec.close()


class _LateBound:
    def __init__(self, owner):
        self._owner = owner

    def __getattr__(self, name):
        owner = self._owner
        attr = getattr(owner._current(), name)
        if not callable(attr):
            return attr
        return lambda *a, **k: getattr(owner._current(), name)(*a, **k)


def _variant_obstacles(version):
    return get_variant(version).obstacle_map() != 0


def _default_variant(rows, columns, obstacle_map=None):
    """The reference's StrategoProceduralEnv is built from (rows, columns) only.  For a board size one of the reference's variants
    has, that variant with the most pieces (its capture-event capacity covers the others), else a custom variant of that size;
    the obstacle cells are the caller's (they are a per-handle constant here, taken from the first state / obstacle map seen)."""
    import dataclasses
    from .config import VARIANTS, custom_variant
    cands = [v for v in VARIANTS.values() if v.rows == rows and v.columns == columns]
    v = max(cands, key=lambda v: sum(v.piece_counts)) if cands else custom_variant(rows, columns)
    if obstacle_map is not None:
        obst = tuple((int(r), int(c)) for r, c in zip(*np.nonzero(np.asarray(obstacle_map))))
        if obst != tuple(sorted(v.obstacle_locations)):
            v = dataclasses.replace(v, name='%s_obstacles_%x' % (v.name, hash(obst) & 0xFFFFFFFF), obstacle_locations=obst, human_inits='')
    return v


class StrategoProceduralEnv:
    """The reference class of the same name (penv:20-214) for ONE state at a time: same constructor, method names,
    argument meaning, return types (numpy int64 states and masks, float32 observations, Python scalars) and errors.  It is
    what `StrategoMultiAgentEnv.base_env` is; every call runs the HIP kernels through a one-game
    BatchedStrategoProceduralEnv (use that class directly for throughput)."""

    def __init__(self, rows, columns, version=None, device=0):
        if rows < 3 or columns < 3:
            raise ValueError("Both rows and columns have to be at least 3 (you passed rows: {} columns: {})."
                             .format(rows, columns))                                                # penv:28-30
        self.rows, self.columns = np.int64(rows), np.int64(columns)
        self.action_size = np.int64(ia.action_size(int(rows), int(columns)))                        # penv:34
        k = ia.spatial_channels(int(rows), int(columns))
        self.spatial_action_size = (np.int64(rows), np.int64(columns), np.int64(k))                 # penv:35
        if int(rows) * int(columns) > 256:
            raise ValueError("boards of more than 256 cells are not supported ({} x {})".format(rows, columns))
        self._version = version
        self._device = device
        self._by_obstacles = {}      # obstacle map bytes -> one-state batched env (obstacles are a per-handle constant)
        self._batched = None

    def _for_obstacles(self, obstacle_map):
        """The one-state batched env whose handle carries these obstacle cells (the reference reads them from every state)."""
        ob = np.ascontiguousarray(np.asarray(obstacle_map) != 0)
        key = ob.tobytes()
        b = self._by_obstacles.get(key)
        if b is None:
            v = self._version if self._version is not None else _default_variant(int(self.rows), int(self.columns), ob)
            b = BatchedStrategoProceduralEnv(v, 1, device=self._device)
            b.strict = True          # a state the packed record cannot carry is an error here, never silently altered
            self._by_obstacles[key] = b
        self._batched = b
        return b

    @property
    def _b(self):
        """The one-state env the call goes to.  Methods are written `self._b.method(self._state(state), ...)`: Python evaluates
        `self._b` BEFORE the arguments, and `_state()` is what selects the env (by the state's obstacle layer) -- so this returns
        a proxy that picks the env when the method is finally called."""
        return _LateBound(self)

    def _current(self):
        if self._batched is None:
            self._for_obstacles(np.zeros((int(self.rows), int(self.columns)), dtype=bool) if self._version is None
                                else _variant_obstacles(self._version))
        return self._batched

    def _state(self, state):
        st = np.asarray(state, dtype=np.int64)
        if st.shape != (NUM_STATE_LAYERS, int(self.rows), int(self.columns)):
            raise ValueError("state must have shape (34, rows, columns)")
        if self._version is None:
            self._for_obstacles(st[2])
        if not np.array_equal(st[2] != 0, self._b.variant.obstacle_map() != 0):
            raise ValueError("the state's obstacle layer differs from the %s variant's (a per-handle constant here)" % self._version)
        return st[None]

    @staticmethod
    def _pl(player):
        return np.asarray([int(player)], dtype=np.int8)

    # ---- state construction / transition ---------------------------------------------------------------------------
    def create_initial_state(self, obstacle_map, player_1_initial_piece_map, player_2_initial_piece_map, max_turns):
        correct_shape = (int(self.rows), int(self.columns))
        for name, m in (("obstacle map", obstacle_map), ("player_1_initial_piece_map map", player_1_initial_piece_map),
                        ("player_2_initial_piece_map map", player_2_initial_piece_map)):
            if tuple(np.shape(m)) != correct_shape:
                raise ValueError("{} needs to be of shape {}, was {}".format(name, correct_shape, np.shape(m)))   # penv:44-55
        if self._version is None:
            self._for_obstacles(obstacle_map)
        elif not np.array_equal(np.asarray(obstacle_map) != 0, self._b.variant.obstacle_map() != 0):
            raise ValueError("obstacle_map differs from the %s variant's (a per-handle constant here)" % self._version)
        st = self._b.create_initial_state(np.asarray(player_1_initial_piece_map, dtype=np.int8)[None],
                                          np.asarray(player_2_initial_piece_map, dtype=np.int8)[None])[0].cpu().numpy()
        st[5, 1, 0] = int(max_turns)                                                                # StateData.MAX_TURNS (impl:247)
        return st

    def get_next_state(self, state, player, action_index, allow_piece_oscillation=False):          # penv:148-155
        ns, _, ok = self._b.get_next_state(self._state(state), self._pl(player), np.asarray([int(action_index)], dtype=np.int64),
                                           allow_piece_oscillation=allow_piece_oscillation)
        if not bool(ok[0]):
            raise ValueError("Couldn't get the next state because the move wasn't valid.")          # impl:902
        return ns[0].cpu().numpy(), player * -1

    def is_move_valid_by_position(self, state, player, start_r, start_c, end_r, end_c, allow_piece_oscillation=False):
        return bool(self._b.is_move_valid_by_position(self._state(state), self._pl(player), [int(start_r)], [int(start_c)],
                                                      [int(end_r)], [int(end_c)], allow_piece_oscillation)[0])

    def is_move_valid_by_1d_index(self, state, player, action_index, allow_piece_oscillation=False):
        return bool(self._b.is_move_valid_by_1d_index(self._state(state), self._pl(player), [int(action_index)],
                                                      allow_piece_oscillation)[0])

    # ---- masks -------------------------------------------------------------------------------------------------------
    def get_valid_moves_as_1d_mask(self, state, player, player_perspective=False):                  # penv:74-80
        m = self._b.get_valid_moves_as_1d_mask(self._state(state), self._pl(player), player_perspective=player_perspective)
        return m[0].cpu().numpy().astype(np.int64)

    def get_valid_moves_as_spatial_mask(self, state, player):                                       # penv:127-128
        return self._b.get_valid_moves_as_spatial_mask(self._state(state), self._pl(player))[0].cpu().numpy().astype(np.int64)

    def get_dict_of_valid_moves_by_position(self, state, player):                                   # penv:82-85
        return self._b.get_dict_of_valid_moves_by_position(self._state(state), self._pl(player))[0]

    # ---- observations (raw) --------------------------------------------------------------------------------------------
    def get_fully_observable_observation(self, state, player):                                      # penv:157-160
        return self._b.get_fully_observable_observation(self._state(state), self._pl(player))[0].cpu().numpy()

    def get_partially_observable_observation(self, state, player):                                  # penv:162-164
        return self._b.get_partially_observable_observation(self._state(state), self._pl(player))[0].cpu().numpy()

    def get_fully_observable_observation_extended_channels(self, state, player):                    # penv:166-169
        return self._b.get_fully_observable_observation_extended_channels(self._state(state), self._pl(player))[0].cpu().numpy()

    def get_partially_observable_observation_extended_channels(self, state, player):                # penv:171-173
        return self._b.get_partially_observable_observation_extended_channels(self._state(state), self._pl(player))[0].cpu().numpy()

    def get_serializable_string_for_fully_observable_state(self, state):                            # penv:175-177
        return self._b.get_serializable_string_for_fully_observable_state(self._state(state))[0]

    def get_serializable_string_for_partially_observable_state(self, state):                        # penv:179-181
        return self._b.get_serializable_string_for_partially_observable_state(self._state(state))[0]

    def print_board_to_console(self, state, partially_observable=False, hide_still_piece_markers=True):   # penv:183-214
        BatchedStrategoProceduralEnv.print_board_to_console(state, partially_observable, hide_still_piece_markers)

    # ---- state algebra -------------------------------------------------------------------------------------------------
    def get_state_from_player_perspective(self, state, player):                                     # penv:101-103
        st = np.asarray(state, dtype=np.int64)
        return self._b.get_state_from_player_perspective(st[None], self._pl(player))[0].cpu().numpy()

    def get_game_ended(self, state, player):                                                        # penv:141-143
        return np.float32(self._b.get_game_ended(np.asarray(state, dtype=np.int64)[None], self._pl(player))[0].item())

    def get_game_result_is_invalid(self, state):                                                    # penv:145-146
        return bool(self._b.get_game_result_is_invalid(np.asarray(state, dtype=np.int64)[None])[0])

    # ---- index converters (scalars, like the reference) ---------------------------------------------------------------
    def get_action_1d_index_from_positions(self, start_r, start_c, end_r, end_c):                   # penv:62-66
        return np.int64(self._b.get_action_1d_index_from_positions(int(start_r), int(start_c), int(end_r), int(end_c)))

    def get_action_positions_from_1d_index(self, action_index):                                     # penv:68-72
        return tuple(np.int64(x) for x in self._b.get_action_positions_from_1d_index(int(action_index)))

    def get_action_positions_from_player_perspective(self, player, start_r, start_c, end_r, end_c):   # penv:105-108
        return tuple(np.int64(x) for x in self._b.get_action_positions_from_player_perspective(
            int(player), int(start_r), int(start_c), int(end_r), int(end_c)))

    def get_action_1d_index_from_player_perspective(self, action_index, player):                    # penv:110-115
        return np.int64(self._b.get_action_1d_index_from_player_perspective(int(action_index), int(player)))

    def get_action_spatial_index_from_positions(self, start_r, start_c, end_r, end_c):              # penv:117-121
        return tuple(np.int64(x) for x in self._b.get_action_spatial_index_from_positions(int(start_r), int(start_c),
                                                                                          int(end_r), int(end_c)))

    def get_action_positions_from_spatial_index(self, spatial_index):                               # penv:123-125
        r, c, ch = (int(x) for x in spatial_index)
        return tuple(np.int64(x) for x in self._b.get_action_positions_from_spatial_index((r, c, ch)))

    def get_action_1d_index_from_spatial_index(self, spatial_index):                                # penv:130-133
        r, c, ch = (int(x) for x in spatial_index)
        return np.int64(self._b.get_action_1d_index_from_spatial_index((r, c, ch)))

    def get_action_spatial_index_from_1d_index(self, action_index):                                 # penv:135-139
        return tuple(np.int64(x) for x in self._b.get_action_spatial_index_from_1d_index(int(action_index)))

    def close(self):
        for b in self._by_obstacles.values():
            b.close()
        self._by_obstacles = {}
        self._batched = None
