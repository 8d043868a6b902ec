"""Goldens for obs_channel_mode='original' (deprecated 32/33-layer observations, maenv:368-375), generated from the
REFERENCE (BUILD CONTAINER ONLY) in observation_mode=BOTH_OBSERVATIONS.  Same digest convention as gen_golden_both.py.
Output: tests/golden/games_orig_<variant>.npz and tests/golden/orig_norm.json (the reference's mids / ranges).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle import gen_golden as G  # noqa: E402
from tools.oracle import gen_golden_both as B  # noqa: E402

if __name__ == '__main__':
    norm = B.generate(channel_mode='original', prefix='games_orig',
                      plan=(('barrage', 8), ('standard', 1), ('tiny', 16), ('micro', 16), ('fives', 8), ('octa_barrage', 4)),
                      seed_offset=7000)
    with open(os.path.join(G.GOLD, 'orig_norm.json'), 'w') as f:
        json.dump(norm, f)
