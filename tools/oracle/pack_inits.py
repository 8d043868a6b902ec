"""Pack the reference's Gravon human-setup lists into binary tables (BUILD CONTAINER ONLY).

Reads BARRAGE_INITS / STANDARD_INITS (reference game/inits/*.py: lists of 40-char strings over
'A'..'M', legend util.py:84-112 / convert_letter_to_num_left util.py:154-180) and writes
stratego_env_amd/inits/<name>_setups.npy: uint8 [n, 20], two piece codes per byte (low nibble =
even string position), list order and duplicates preserved so that index i means HUMAN_INITS[i].
The product decodes them in stratego_env_amd/setups.py; this script also verifies that decode
against the reference's create_initial_positions_from_human_data (util.py:241-275).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402

LETTER_TO_CODE = {'A': 0, 'B': 12, 'C': 1, 'D': 2, 'E': 3, 'F': 4, 'G': 5, 'H': 6, 'I': 7, 'J': 8, 'K': 9, 'L': 10, 'M': 11}


def main():
    ref = import_reference()
    from stratego_env.game.inits.barrage_human_inits import BARRAGE_INITS
    from stratego_env.game.inits.standard_human_inits import STANDARD_INITS
    from stratego_env_amd import setups as S
    out_dir = os.path.join(ROOT, 'stratego_env_amd', 'inits')
    os.makedirs(out_dir, exist_ok=True)
    for name, lst in (('barrage', BARRAGE_INITS), ('standard', STANDARD_INITS)):
        codes = np.zeros((len(lst), 40), dtype=np.uint8)
        for i, s in enumerate(lst):
            assert len(s) == 40
            codes[i] = [LETTER_TO_CODE[ch] for ch in s]
        packed = (codes[:, 0::2] | (codes[:, 1::2] << 4)).astype(np.uint8)
        np.save(os.path.join(out_dir, name + '_setups.npy'), packed)
        table = S.load_setup_table(name)
        assert np.array_equal(table, codes)
        # verify the own-side map decode against the reference for a sample of string pairs
        cfg = ref.config.BARRAGE_STRATEGO_CONFIG if name == 'barrage' else ref.config.STANDARD_STRATEGO_CONFIG
        rs = np.random.RandomState(0)
        idx = list(range(8)) + list(range(len(lst) - 8, len(lst))) + list(rs.randint(0, len(lst), 48))
        for a, b in zip(idx, idx[::-1]):
            pos = ref.util.create_initial_positions_from_human_data(lst[a], lst[b], cfg)
            m1, m2 = S.own_side_maps(table[a], table[b], 10, 10, 4)
            assert np.array_equal(pos[0], m1) and np.array_equal(pos[1], m2), (name, a, b)
        print(name, packed.shape, 'ok')


if __name__ == '__main__':
    main()
