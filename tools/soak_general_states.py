"""Soak of the general-state pass (sgx_step_states' second pass) against the oracle: fresh random states of legal
values and impossible structure (tests.helpers.general_states) every round, on every compiled-in board size -- masks in both encodings,
all four raw observation kinds, next state / validity for valid and garbage 1-D actions with and without the oscillation flag, validity
by position -- for a wall-clock budget.  No state may come back flagged.

    python tools/soak_general_states.py [seconds=120]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from stratego_env_amd.config import VARIANTS  # noqa: E402
from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv  # noqa: E402
from tests.helpers import general_states  # noqa: E402

PLAN = [('barrage', 24), ('medium', 32), ('octa_barrage', 24), ('standard2', 6), ('fives', 32), ('tiny', 32), ('micro', 32)]
OBS = ('get_partially_observable_observation_extended_channels', 'get_fully_observable_observation_extended_channels',
       'get_partially_observable_observation', 'get_fully_observable_observation')


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t0, rounds, checked = time.time(), 0, 0
    envs = {name: BatchedStrategoProceduralEnv(name, n) for name, n in PLAN}
    while time.time() - t0 < budget:
        rs = np.random.RandomState(1000 + rounds)
        for name, n in PLAN:
            v = VARIANTS[name]
            pe, ru = envs[name], orc.OracleRules(v.rows, v.columns)
            states, players = general_states(name, n, rs)
            m1 = pe.get_valid_moves_as_1d_mask(states, players).cpu().numpy()
            assert int(pe.last_sanitised.sum()) == 0, (name, rounds, '1d mask flagged')
            ms = pe.get_valid_moves_as_spatial_mask(states, players).cpu().numpy()
            obs = []
            for fn in OBS:
                obs.append(getattr(pe, fn)(states, players).cpu().numpy())
                assert int(pe.last_sanitised.sum()) == 0, (name, rounds, fn)
            osc = bool(rs.randint(2))
            acts = np.asarray([int(rs.choice(np.flatnonzero(m1[e]))) if rs.rand() < 0.7 else int(rs.randint(-3, ru.action_size + 3))
                               for e in range(n)], dtype=np.int64)
            ns, npl, ok = pe.get_next_state(states, players, acts, allow_piece_oscillation=osc)
            assert int(pe.last_sanitised.sum()) == 0, (name, rounds, 'get_next_state flagged')
            ns, npl, ok = ns.cpu().numpy(), npl.cpu().numpy(), ok.cpu().numpy()
            pos = rs.randint(-1, max(v.rows, v.columns) + 1, size=(n, 4))
            byp = pe.is_move_valid_by_position(states, players, pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3]).cpu().numpy()
            for e in range(n):
                st, pl = states[e], int(players[e])
                assert np.array_equal(m1[e], ru.get_valid_moves_as_1d_mask(st, pl)), (name, rounds, e, '1d mask')
                assert np.array_equal(ms[e], ru.get_valid_moves_as_spatial_mask(st, pl)), (name, rounds, e, 'spatial mask')
                for fn, o in zip(OBS, obs):
                    assert o[e].tobytes() == getattr(ru, fn)(st, pl).tobytes(), (name, rounds, e, fn)
                want = ru.is_move_valid_by_1d_index(st, pl, int(acts[e]), allow_piece_oscillation=osc)
                assert bool(ok[e]) == want, (name, rounds, e, int(acts[e]), osc)
                if want:
                    w, wp = ru.get_next_state(st, pl, int(acts[e]), allow_piece_oscillation=osc)
                    assert np.array_equal(ns[e], w) and npl[e] == wp, (name, rounds, e, 'next state')
                else:
                    assert np.array_equal(ns[e], st) and npl[e] == pl
                assert bool(byp[e]) == ru.is_move_valid_by_position(st, pl, *[int(x) for x in pos[e]]), (name, rounds, e, 'by position')
            checked += n
        rounds += 1
    print("general-state soak ok: %d rounds, %d states (x 8 functions) checked against the oracle in %.0f s, none flagged" % (rounds, checked, time.time() - t0))


if __name__ == '__main__':
    main()
