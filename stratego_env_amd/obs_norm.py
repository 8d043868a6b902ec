"""Per-channel observation bounds and the [-1, 1] normalisation constants of StrategoMultiAgentEnv (host side).

The HIP kernels emit observations already normalised (through the look-up tables of sgx_build_*_lut); these numpy
constants exist for the facade's reference attributes (`_p_obs_highs`, `_p_obs_mids`, ...) and its
`normalize_* / denormalize_*` helpers (maenv:499-511).

piece_counts: counts of piece codes 1..12 (SPY .. BOMB), as in config.Variant.
"""
import numpy as np

RECENT_HI, RECENT_LO = 1.0, -3.0   # RecentMoves.JUST_CAME_FROM / JUST_ARRIVED_AND_CANT_DOUBLE_BACK (impl:28-32)


def obs_highs_lows(piece_counts, full, original):
    """(highs, lows) float32 [channels].

    extended: maenv:261-313 (partial, 67) / maenv:202-258 (full, 79); original: maenv:146-199 (32) / maenv:87-143 (33)."""
    if not original:
        n_true = 24 if full else 12
        po_end = n_true + 26
        n = po_end + 3 + 24 + 2
        hi, lo = np.ones(n, dtype=np.float32), -np.ones(n, dtype=np.float32)
        rec0, cap0 = po_end + 1, po_end + 3
        hi[cap0:cap0 + 24], lo[cap0:cap0 + 24] = 8.0, 0.0
    else:
        n = 33 if full else 32
        hi, lo = np.full(n, 2.0, dtype=np.float32), np.zeros(n, dtype=np.float32)
        if full:
            hi[0:2], hi[5:7] = 12.0, 13.0          # SP.BOMB, SP.UNKNOWN
            rec0, cap0 = 3, 7
        else:
            hi[0], hi[1:3] = 12.0, 13.0
            rec0, cap0 = 4, 6
    hi[rec0:rec0 + 2], lo[rec0:rec0 + 2] = RECENT_HI, RECENT_LO
    for t in range(1, 13):
        if piece_counts[t - 1] > 1:
            hi[cap0 + t - 1] = hi[cap0 + 12 + t - 1] = piece_counts[t - 1]
    return hi, lo


def ranges_mids(highs, lows):
    """maenv:388-396: float32, shaped (1, 1, channels)."""
    n = highs.shape[0]
    ranges = np.reshape((highs - lows) / np.float32(2.0), (1, 1, n))
    mids = np.reshape((highs + lows) / np.float32(2.0), (1, 1, n))
    return ranges, mids
