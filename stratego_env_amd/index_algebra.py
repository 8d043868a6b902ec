"""Action-index algebra of the reference (stratego_procedural_impl.py:166-169, 252-396, 678-720) on numpy integer arrays.

Bijections between the flat spatial index a in [0, R*C*K), (r, c, channel), (start_r, start_c, end_r, end_c) and the 1-D
index ((sr*C + sc) * (R+C)) + (end_r if the row changed else R + end_c), plus the 180-degree flip for player -1.
Python floor-division semantics are kept for out-of-board values, like the reference's integer code.
"""
import numpy as np


def spatial_channels(R, C):
    return 2 * (R - 1) + 2 * (C - 1) + 1                       # impl:257-259


def action_size(R, C):
    return R * C * (R + C) + 1                                  # impl:252-254


def action_1d_from_positions(R, C, sr, sc, er, ec):            # impl:262-277
    sr, sc, er, ec = (np.asarray(x, dtype=np.int64) for x in (sr, sc, er, ec))
    off = np.where(er != sr, er, R + ec)
    return (sr * C + sc) * (R + C) + off


def positions_from_spatial(R, C, r, c, ch):                     # impl:314-335
    r, c, ch = (np.asarray(x, dtype=np.int64) for x in (r, c, ch))
    mr, mc = R - 1, C - 1
    er = np.where(ch < mr, r + ch + 1, np.where(ch < 2 * mr, r - (ch - mr + 1), r))
    ec = np.where(ch < 2 * mr, c, np.where(ch < 2 * mr + mc, c + (ch - 2 * mr + 1), c - (ch - (2 * mr + mc) + 1)))
    return r, c, er, ec


def action_1d_from_spatial(R, C, r, c, ch):                     # impl:338-347
    return action_1d_from_positions(R, C, *positions_from_spatial(R, C, r, c, ch))


def positions_from_1d(R, C, idx):                               # impl:350-383 (the no-op index is the caller's business)
    idx = np.asarray(idx, dtype=np.int64)
    mpa = R + C
    q = idx // mpa
    sr, sc, off = q // C, q % C, idx % mpa
    er = np.where(off >= R, sr, off)
    ec = np.where(off >= R, off - R, sc)
    return sr, sc, er, ec


def spatial_from_positions(R, C, sr, sc, er, ec):               # impl:280-311; channel -1 where no straight non-null move
    sr, sc, er, ec = (np.asarray(x, dtype=np.int64) for x in (sr, sc, er, ec))
    dr, dc = er - sr, ec - sc
    off = np.where(dr > 0, 0, np.where(dr < 0, R - 1, np.where(dc > 0, 2 * (R - 1), 2 * (R - 1) + (C - 1))))
    ch = off + np.abs(dr + dc) - 1
    bad = ((dr != 0) & (dc != 0)) | ((dr == 0) & (dc == 0))
    return sr, sc, np.where(bad, -1, ch)


def flip_positions(R, C, sr, sc, er, ec):                       # impl:678-695 (player -1)
    return R - 1 - sr, C - 1 - sc, R - 1 - er, C - 1 - ec


def action_1d_from_player_perspective(R, C, idx, player):       # impl:698-720
    idx = np.asarray(idx, dtype=np.int64)
    if player == 1:
        return idx
    flipped = action_1d_from_positions(R, C, *flip_positions(R, C, *positions_from_1d(R, C, idx)))
    return np.where(idx == action_size(R, C) - 1, idx, flipped)
