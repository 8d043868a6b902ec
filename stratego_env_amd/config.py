"""Game-variant tables (reference: stratego_env/game/config.py:3-313).

Each variant is a `Variant` with the same information the reference keeps in its `*_STRATEGO_CONFIG`
dicts: board size, max_turns, obstacle cells, pieces per side by piece code, usable back rows.
`tests/test_config.py` checks every field against `tests/golden/variants.json`, which
`tools/oracle/gen_golden.py` dumps from the imported reference.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .enums import GameVersions

_LAKES_10 = ((4, 2), (5, 2), (4, 3), (5, 3), (4, 6), (5, 6), (4, 7), (5, 7))

#                      spy scout miner sgt lt cpt maj col gen mar flag bomb   (piece codes 1..12)
_STANDARD_PIECES = (1, 8, 5, 4, 4, 4, 3, 2, 1, 1, 1, 6)
_BARRAGE_PIECES = (1, 2, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1)

NUM_PIECE_TYPES = 12
PO_OBS_CHANNELS = 67   # reference impl:1332
FO_OBS_CHANNELS = 79   # reference impl:1227
PO_OBS_CHANNELS_ORIGINAL = 32   # reference impl:1148 (obs_channel_mode='original')
FO_OBS_CHANNELS_ORIGINAL = 33   # reference impl:1070
NUM_STATE_LAYERS = 34  # reference impl:109


@dataclass(frozen=True)
class Variant:
    name: str
    rows: int
    columns: int
    max_turns: int
    obstacle_locations: Tuple[Tuple[int, int], ...]
    piece_counts: Tuple[int, ...]          # index t-1 holds the count of piece code t (1..12)
    initial_state_usable_rows: int
    human_inits: str = ''                  # name of the packed Gravon table, '' if unsupported (util.py:305-310)
    capture_capacity: int = 0              # most pieces one side may have on the board when that exceeds sum(piece_counts): an
    #                                        env_config that overrides `piece_amounts` changes the normalisation only (maenv:323-326,
    #                                        370-382), the setups keep the version's pieces; sizes the capture-event list

    @property
    def cells(self) -> int:
        return self.rows * self.columns

    @property
    def spatial_channels(self) -> int:
        """K = ways to move = 2(R-1) + 2(C-1) + 1 (reference impl:257-259)."""
        return 2 * (self.rows - 1) + 2 * (self.columns - 1) + 1

    @property
    def spatial_action_size(self) -> Tuple[int, int, int]:
        return (self.rows, self.columns, self.spatial_channels)

    @property
    def num_spatial_actions(self) -> int:
        return self.cells * self.spatial_channels

    @property
    def action_size(self) -> int:
        """1-D action encoding size R*C*(R+C)+1 (reference impl:252-254)."""
        return self.cells * (self.rows + self.columns) + 1

    @property
    def pieces_per_side(self) -> int:
        return int(sum(self.piece_counts))

    @property
    def max_pieces_on_board(self) -> int:
        return max(self.pieces_per_side, int(self.capture_capacity))

    def piece_amounts(self) -> Dict[int, int]:
        """{piece code: count}, the reference's `piece_amounts` keyed by code instead of the SP enum."""
        return {t + 1: n for t, n in enumerate(self.piece_counts)}

    def obstacle_map(self) -> np.ndarray:
        m = np.zeros((self.rows, self.columns), dtype=np.uint8)
        for r, c in self.obstacle_locations:
            m[r, c] = 1
        return m

    def as_reference_config(self) -> dict:
        """The reference's `*_STRATEGO_CONFIG` dict for this variant (game/config.py:3-313): same keys, `piece_amounts`
        keyed by the SP enum in SPY..BOMB order, obstacle cells as a list of (row, column) tuples."""
        from .enums import SP
        return {'rows': self.rows, 'columns': self.columns, 'max_turns': self.max_turns,
                'obstacle_locations': [tuple(x) for x in self.obstacle_locations],
                'piece_amounts': {SP(t): n for t, n in enumerate(self.piece_counts, start=1)},
                'initial_state_usable_rows': self.initial_state_usable_rows}

    def captured_count_highs(self) -> Tuple[int, ...]:
        """Normalisation highs of the captured-count channels: count if > 1 else 8 (maenv:288-298)."""
        return tuple(n if n > 1 else 8 for n in self.piece_counts)


def _v(name, r, c, turns, obst, pieces, usable, inits=''):
    return Variant(name, r, c, turns, tuple(obst), tuple(pieces), usable, inits)


VARIANTS: Dict[str, Variant] = {
    'standard': _v('standard', 10, 10, 2000, _LAKES_10, _STANDARD_PIECES, 4, 'standard'),
    'medium_standard': _v('medium_standard', 10, 10, 800, _LAKES_10, _STANDARD_PIECES, 4, 'standard'),
    'short_standard': _v('short_standard', 10, 10, 400, _LAKES_10, _STANDARD_PIECES, 4, 'standard'),
    'standard2': _v('standard2', 15, 15, 2000, (), (0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 1, 0), 5),
    'barrage': _v('barrage', 10, 10, 1000, _LAKES_10, _BARRAGE_PIECES, 4, 'barrage'),
    'short_barrage': _v('short_barrage', 10, 10, 100, _LAKES_10, _BARRAGE_PIECES, 4, 'barrage'),
    'octa_barrage': _v('octa_barrage', 8, 8, 1000, ((4, 2), (3, 2), (4, 5), (3, 5)), _BARRAGE_PIECES, 3),
    'medium': _v('medium', 6, 6, 200, (), (0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0), 1),
    'fives': _v('fives', 5, 5, 60, (), (0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0), 1),
    'tiny': _v('tiny', 4, 4, 100, (), (0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 0), 1),
    'micro': _v('micro', 3, 4, 20, (), (0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0), 1),
}


MAX_PIECES_PER_TYPE = 127     # SGX_MAX_PIECES_PER_TYPE (a capture event counts to 8; more pieces of a type chain events)


def custom_variant(rows, columns, max_turns=2000, obstacle_locations=(), piece_counts=None, initial_state_usable_rows=None,
                   name=None, human_inits='', capture_capacity=0) -> Variant:
    """A variant the reference has no name for: any board of rows, columns >= 3 (the reference's StrategoProceduralEnv takes any
    size, penv:27-36).  Without piece_counts the side gets one of each movable rank that fits plus a flag -- the count only sizes
    the capture-event list and the random setups; the functional API works on the caller's states."""
    rows, columns = int(rows), int(columns)
    if rows < 3 or columns < 3:
        raise ValueError("Both rows and columns have to be at least 3 (you passed rows: {} columns: {}).".format(rows, columns))
    usable = int(initial_state_usable_rows) if initial_state_usable_rows is not None else max(1, (rows - 1) // 2)
    if piece_counts is None:
        room = usable * columns
        piece_counts = [0] * NUM_PIECE_TYPES
        piece_counts[10] = 1                                   # flag
        for t in (1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 11):           # scout .. marshal, spy, bomb; then more scouts
            if sum(piece_counts) < room:
                piece_counts[t] = 1
        # more scouts, at most 8 of them like Standard: the remaining setup cells stay empty
        piece_counts[1] += min(room - sum(piece_counts), 8 - piece_counts[1])
    if max(piece_counts) > MAX_PIECES_PER_TYPE:
        raise ValueError("at most %d pieces of one type per side (got %s)" % (MAX_PIECES_PER_TYPE, list(piece_counts)))
    return Variant(name or 'custom_%dx%d' % (rows, columns), rows, columns, int(max_turns), tuple(tuple(x) for x in obstacle_locations),
                   tuple(int(x) for x in piece_counts), usable, human_inits, int(capture_capacity))


def piece_counts_from_amounts(piece_amounts) -> Tuple[int, ...]:
    """The reference's `piece_amounts` dict (keys: SP members, piece codes or SP names) -> counts of piece codes 1..12."""
    from .enums import SP
    counts = [0] * NUM_PIECE_TYPES
    for k, n in dict(piece_amounts).items():
        code = k.value if isinstance(k, SP) else (SP[k].value if isinstance(k, str) else int(getattr(k, 'value', k)))
        if not 1 <= code <= NUM_PIECE_TYPES:
            raise ValueError("piece_amounts key %r is not a piece" % (k,))
        counts[code - 1] = int(n)
    return tuple(counts)


def get_variant(version) -> Variant:
    """Accepts a GameVersions member, its string value, or a Variant (custom_variant)."""
    if isinstance(version, Variant):
        return version
    if isinstance(version, GameVersions):
        version = version.value
    if version not in VARIANTS:
        raise ValueError("unknown game version {!r}".format(version))
    return VARIANTS[version]


# The reference's module-level names (game/config.py, stratego_multiagent_env.py:33-45)
VERSION_CONFIGS = {GameVersions(name): v.as_reference_config() for name, v in VARIANTS.items()}
STANDARD_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.STANDARD]
MEDIUM_STANDARD_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.MEDIUM_STANDARD]
SHORT_STANDARD_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.SHORT_STANDARD]
STANDARD_STRATEGO_CONFIG2 = VERSION_CONFIGS[GameVersions.STANDARD2]
BARRAGE_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.BARRAGE]
SHORT_BARRAGE_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.SHORT_BARRAGE]
OCTA_BARRAGE_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.OCTA_BARRAGE]
MEDIUM_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.MEDIUM]
FIVES_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.FIVES]
TINY_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.TINY]
MICRO_STRATEGO_CONFIG = VERSION_CONFIGS[GameVersions.MICRO]
