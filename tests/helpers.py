"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import hashlib
import json
import os

import numpy as np

from oracle import oracle as orc
from stratego_env_amd.config import VARIANTS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_variants_json():
    return json.load(open(os.path.join(GOLDEN, 'variants.json')))


def load_kat():
    return json.load(open(os.path.join(GOLDEN, 'kat.json')))


def _load_npz(path):
    # materialise once: NpzFile re-decompresses a member on every [] access
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def load_games(name):
    return _load_npz(os.path.join(GOLDEN, 'games_%s.npz' % name))


def load_expanded(name):
    return _load_npz(os.path.join(GOLDEN, 'expanded_%s.npz' % name))


def oracle_env(name, penalize_ties=False):
    v = VARIANTS[name]
    return orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts, penalize_ties)


def oracle_cvariant(name, setups=None):
    v = VARIANTS[name]
    return orc.make_cvariant(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts,
                             v.initial_state_usable_rows, setups=setups)


def digest_obs(obs):
    """Same digest as tools/oracle/gen_golden.py: sha256 over players ascending of mask(u8) + obs bytes, first 8 bytes."""
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(np.ascontiguousarray(obs[p]['valid_actions_mask']).astype(np.uint8).tobytes())
        h.update(np.ascontiguousarray(obs[p]['partial_observation'], dtype=np.float32).tobytes())
    return int.from_bytes(h.digest()[:8], 'little')
