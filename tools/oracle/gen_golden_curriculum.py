"""Goldens for the curriculum-start-state reset path (maenv:341-346, 519-527; util.py:322-387) and for the operator-level
debugging aids (print_board_to_console penv:183-214, get_dict_of_valid_moves_by_position penv:82-85, the serializable
strings penv:175-181), generated from the REFERENCE (BUILD CONTAINER ONLY).

h5py is not installed here, so the reference's `h5py.File` is pointed at an in-memory stand-in that serves the committed
`tests/golden/curriculum_barrage.npz` arrays ('state' int64 [n,34,10,10], 'winner' int64 [n]) -- reading a file is not
game logic; every other line that runs is the reference's.  Output: tests/golden/curriculum_barrage.npz, curriculum.json,
board_utils.json.
"""
import contextlib
import hashlib
import io
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402
from tools.oracle import gen_golden as G  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from stratego_env_amd import setups as S  # noqa: E402
from stratego_env_amd.config import VARIANTS  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def obs_digest(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(np.asarray(obs[p]['valid_actions_mask']).astype(np.uint8).tobytes())
        h.update(np.ascontiguousarray(obs[p]['partial_observation'], dtype=np.float32).tobytes())
    return h.hexdigest()[:16]


def make_curriculum_states(n=8):
    """Mid-game Barrage positions from oracle rollouts; 'winner' = who eventually won that rollout (+1 / -1)."""
    v = VARIANTS['barrage']
    cv = orc.make_cvariant(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts, v.initial_state_usable_rows,
                           setups=S.load_setup_table('barrage'))
    ru = orc.OracleRules(v.rows, v.columns)
    states, winners = [], []
    g = 0
    while len(states) < n:
        st, pl, t, snap = orc.reset_state(cv, 4242, g, 0), 1, 0, None
        g += 1
        while ru.get_game_ended(st, pl) == 0:
            m = ru.get_valid_moves_as_spatial_mask(ru.get_state_from_player_perspective(st, pl), 1)
            a = orc.sample_action(m.astype(np.uint8), 4242, g, 0, t)
            idx = ru.get_action_1d_index_from_player_perspective(ru.get_action_1d_index_from_spatial_index(
                np.unravel_index(a, m.shape)), pl)
            st, pl = ru.get_next_state(st, pl, idx)
            t += 1
            if t == 60 + 7 * len(states):
                snap = st.copy()
        w = int(st[5, 0, 2])
        if snap is not None and w != 0:
            states.append(snap)
            winners.append(w)
    return np.stack(states), np.asarray(winners, dtype=np.int64)


class _LegacyNumpy:
    """util.py:377 does `np.squeeze((array, offset))`; NumPy < 1.24 (the reference's era) built a ragged object array that
    unpacks back to (array, offset), NumPy 2.x raises.  Everything else is plain numpy."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def squeeze(a, *args, **kw):
        return a if isinstance(a, tuple) else np.squeeze(a, *args, **kw)


class _FakeH5File:
    """Serves the arrays of an .npz the way the reference reads its HDF5 file (util.py:327-370)."""
    data = {}

    def __init__(self, fname, mode='r'):
        pass

    def keys(self):
        return self.data.keys()

    def __getitem__(self, k):
        return self.data[k]

    def close(self):
        pass


def main():
    ref = import_reference()
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    states, winners = make_curriculum_states()
    np.savez_compressed(os.path.join(G.GOLD, 'curriculum_barrage.npz'), state=states, winner=winners)
    _FakeH5File.data = {'state': states, 'winner': winners}
    ref.util.h5py.File = _FakeH5File
    ref.util.h5py.Group = type('Group', (), {})
    ref.util.np = _LegacyNumpy()
    cases = []
    for seed, same in ((0, False), (3, False), (11, False), (5, True)):
        np.random.seed(seed)
        random.seed(seed)
        env = ref.maenv.StrategoMultiAgentEnv({'version': GV.BARRAGE, 'observation_mode': OM.PARTIALLY_OBSERVABLE,
                                               'curriculum_start_states_path': 'curriculum.h5',
                                               'same_start_pos_everytime': same})
        games = []
        for _ in range(3):
            obs = env.reset()
            rec = dict(first_key=int(list(obs.keys())[0]), player=int(env.player), state=sha(env.state), init=obs_digest(obs),
                       steps=[])
            for t in range(12):
                k = list(obs.keys())[0]
                valid = np.flatnonzero(obs[k]['valid_actions_mask'].reshape(-1))
                a = int(valid[(t * 7919) % len(valid)])
                obs, rew, done, info = env.step({k: a})
                rec['steps'].append(dict(action=a, keys=sorted(int(x) for x in obs.keys()), digest=obs_digest(obs),
                                         done=bool(done['__all__']),
                                         rewards={str(kk): float(vv) for kk, vv in rew.items()}))
                if done['__all__']:
                    break
            games.append(rec)
        cases.append(dict(seed=seed, same_start_pos_everytime=same, games=games))
    json.dump(cases, open(os.path.join(G.GOLD, 'curriculum.json'), 'w'))

    # ---- operator-level debugging aids on a few of those states
    pe = ref.penv.StrategoProceduralEnv(10, 10)
    utils = []
    for i in (0, 3, 5):
        st = states[i]
        rec = dict(index=i)
        for po, hide in ((False, True), (True, False)):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                pe.print_board_to_console(st, partially_observable=po, hide_still_piece_markers=hide)
            rec['print_po%d_hide%d' % (po, hide)] = buf.getvalue()
        for pl in (1, -1):
            rec['moves_dict_%d' % pl] = {k: [[int(x) for x in e] for e in val]
                                          for k, val in pe.get_dict_of_valid_moves_by_position(st, pl).items()}
            rec['mask1d_pp_%d' % pl] = sha(np.asarray(pe.get_valid_moves_as_1d_mask(st, pl, player_perspective=True)).astype(np.uint8))
        rec['fo_string_sha'] = hashlib.sha256(pe.get_serializable_string_for_fully_observable_state(st)).hexdigest()[:16]
        rec['po_string_sha'] = hashlib.sha256(pe.get_serializable_string_for_partially_observable_state(st)).hexdigest()[:16]
        utils.append(rec)
    json.dump(utils, open(os.path.join(G.GOLD, 'board_utils.json'), 'w'))
    print(len(cases), 'curriculum cases;', len(utils), 'board-utility records')


if __name__ == '__main__':
    main()
