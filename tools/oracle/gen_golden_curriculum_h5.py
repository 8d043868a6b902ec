"""REAL HDF5 curriculum fixtures and reference-recorded draws from them (BUILD CONTAINER ONLY).

The product interpreter (/usr/bin/python3) has no h5py, but the image carries an Anaconda tree whose python3.9 does
(/opt/conda/bin/python3.9, h5py 3.3.0 on HDF5 1.10): run THIS script with that interpreter.  It

  1. writes the committed curriculum table tests/golden/curriculum_barrage.npz ('state' int64 [n,34,10,10], 'winner' int64 [n]) as
     HDF5 files the way h5py users do -- contiguous datasets, chunked + gzip + shuffle resizable datasets, libver='latest' -- with the
     real library: tests/golden/curriculum_barrage_{contiguous,chunked_gzip,latest}.h5 (fixtures = data, no reference text);
  2. imports the REFERENCE (numba / gym stubbed as everywhere, h5py REAL this time) and lets its own util.load_h5 /
     get_random_curriculum_init_fn (util.py:322-387) read those files after np.random.seed(s): offsets, winners and state digests
     -> tests/golden/curriculum_h5.json;
  3. replays two of the StrategoMultiAgentEnv curriculum episodes of tests/golden/curriculum.json (recorded earlier through a stand-in
     File object) from the real file and insists on the same digests: the stand-in changed nothing.

stratego_env_amd/hdf5_lite.py (a small pure-Python reader for exactly this subset of the format) is tested against these files and draws.

    /opt/conda/bin/python3.9 tools/oracle/gen_golden_curriculum_h5.py
"""
import hashlib
import json
import os
import random
import sys

import h5py  # the real one: must be imported before the stubs are installed
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.oracle.ref_stubs import import_reference  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def obs_digest(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(np.asarray(obs[p]['valid_actions_mask']).astype(np.uint8).tobytes())
        h.update(np.ascontiguousarray(obs[p]['partial_observation'], dtype=np.float32).tobytes())
    return h.hexdigest()[:16]


class _LegacyNumpy:
    """util.py:377 does `np.squeeze((array, offset))`: NumPy < 1.24 built a ragged object array that unpacks back to (array, offset)."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def squeeze(a, *args, **kw):
        return a if isinstance(a, tuple) else np.squeeze(a, *args, **kw)


def write_fixtures(states, winners):
    paths = {}
    p = os.path.join(GOLD, 'curriculum_barrage_contiguous.h5')
    with h5py.File(p, 'w') as f:                                   # what `f.create_dataset(name, data=...)` gives: contiguous layout
        f.create_dataset('state', data=states)
        f.create_dataset('winner', data=winners)
    paths['contiguous'] = p
    p = os.path.join(GOLD, 'curriculum_barrage_chunked_gzip.h5')
    with h5py.File(p, 'w') as f:                                   # a table that was appended to: resizable, chunked, compressed
        d = f.create_dataset('state', shape=(0,) + states.shape[1:], maxshape=(None,) + states.shape[1:], dtype='int64',
                             chunks=(3,) + states.shape[1:], compression='gzip', compression_opts=4, shuffle=True)
        w = f.create_dataset('winner', shape=(0,), maxshape=(None,), dtype='float64', chunks=(5,), fletcher32=True)
        for i in range(len(states)):
            d.resize(i + 1, axis=0); d[i] = states[i]
            w.resize(i + 1, axis=0); w[i] = float(winners[i])
    paths['chunked_gzip'] = p
    p = os.path.join(GOLD, 'curriculum_barrage_latest.h5')
    with h5py.File(p, 'w', libver='latest') as f:                  # the newest file format features: superblock 3, version-2 object headers
        f.create_dataset('state', data=states.astype(np.int16))    # (int16 on disk: a quarter of the bytes; the reference keeps the file's dtype)
        f.create_dataset('winner', data=winners.astype(np.int32))
    paths['latest'] = p
    return paths


def main():
    z = np.load(os.path.join(GOLD, 'curriculum_barrage.npz'))
    states, winners = np.asarray(z['state']), np.asarray(z['winner'])
    paths = write_fixtures(states, winners)
    ref = import_reference()
    assert ref.util.h5py is h5py, "the reference must see the real h5py here"
    ref.util.np = _LegacyNumpy()
    out = {'n': int(len(states)), 'state_sha': sha(states.astype(np.int64)), 'winner': [int(x) for x in winners], 'files': {}}
    for kind, p in paths.items():
        full, off = ref.util.load_h5(fname=p, key='state')
        wfull, _ = ref.util.load_h5(fname=p, key='winner')
        rec = {'file': os.path.basename(p), 'bytes': os.path.getsize(p), 'state_dtype': str(full.dtype), 'winner_dtype': str(wfull.dtype),
               'state_sha_as_int64': sha(np.asarray(full).astype(np.int64)), 'winner_as_int': [int(x) for x in np.asarray(wfull)], 'draws': []}
        assert np.array_equal(np.asarray(full).astype(np.int64), states) and [int(x) for x in np.asarray(wfull)] == [int(x) for x in winners]
        fn = ref.util.get_random_curriculum_init_fn(p, 1000)
        for seed in (0, 1, 2, 3, 11, 12345):
            np.random.seed(seed)
            st, win = fn()
            rec['draws'].append({'seed': seed, 'winner': int(win), 'state_sha_as_int64': sha(np.asarray(st).astype(np.int64)),
                                 'max_turns': int(np.asarray(st)[5, 1, 0]), 'turn': int(np.asarray(st)[5, 0, 0])})
        out['files'][kind] = rec
    # the env-level episodes recorded through the stand-in File object, now from the real file
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    old = json.load(open(os.path.join(GOLD, 'curriculum.json')))
    checked = 0
    for case in old[:2]:
        np.random.seed(case['seed'])
        random.seed(case['seed'])
        env = ref.maenv.StrategoMultiAgentEnv({'version': GV.BARRAGE, 'observation_mode': OM.PARTIALLY_OBSERVABLE,
                                               'curriculum_start_states_path': paths['contiguous'],
                                               'same_start_pos_everytime': case['same_start_pos_everytime']})
        for g in case['games']:
            obs = env.reset()
            assert sha(env.state) == g['state'] and obs_digest(obs) == g['init'], 'the stand-in File object changed something'
            for srec in g['steps']:
                k = list(obs.keys())[0]
                obs, rew, done, info = env.step({k: srec['action']})
                assert obs_digest(obs) == srec['digest']
                checked += 1
    out['env_episodes_rechecked_from_the_real_file'] = checked
    json.dump(out, open(os.path.join(GOLD, 'curriculum_h5.json'), 'w'), indent=1)
    print({k: v['bytes'] for k, v in out['files'].items()}, 'env steps rechecked:', checked)


if __name__ == '__main__':
    main()
