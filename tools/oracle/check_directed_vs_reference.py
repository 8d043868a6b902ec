"""Oracle vs the imported reference (BUILD CONTAINER ONLY) on the constructed positions of tests.helpers.directed_positions
(the SURVEY A.8 quirks: full combat matrix for both players, last-turn flag capture, scout reveals, the stuck mover's no-op):
next state, validity, game result and all four observation kinds."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.oracle.ref_stubs import import_reference  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import directed_positions  # noqa: E402


def main():
    ref = import_reference()
    pe, ru = ref.penv.StrategoProceduralEnv(4, 4), orc.OracleRules(4, 4)
    states, players, actions = directed_positions()
    for i in range(len(states)):
        st, pl, a = states[i], int(players[i]), int(actions[i])
        assert bool(pe.is_move_valid_by_1d_index(st, pl, a)) and ru.is_move_valid_by_1d_index(st, pl, a), i
        want, wpl = pe.get_next_state(st, pl, a)
        got, gpl = ru.get_next_state(st, pl, a)
        assert np.array_equal(want, got) and wpl == gpl, i
        # (the reference declares a float32 return; without Numba the tie value 1e-4 stays a Python float -- SURVEY 8c)
        assert np.float32(pe.get_game_ended(want, wpl)) == np.float32(ru.get_game_ended(got, gpl)), i
        assert bool(pe.get_game_result_is_invalid(want)) == ru.get_game_result_is_invalid(got), i
        for fn in ('get_partially_observable_observation_extended_channels', 'get_fully_observable_observation_extended_channels',
                   'get_partially_observable_observation', 'get_fully_observable_observation'):
            assert getattr(pe, fn)(want, wpl).tobytes() == getattr(ru, fn)(got, gpl).tobytes(), (i, fn)
        assert np.array_equal(pe.get_valid_moves_as_1d_mask(want, wpl), ru.get_valid_moves_as_1d_mask(got, gpl)), i
    print("oracle == reference on %d constructed positions" % len(states))


if __name__ == '__main__':
    main()
