"""Goldens for observation_mode=BOTH_OBSERVATIONS (the reference's DEFAULT_CONFIG mode), generated from the REFERENCE
(BUILD CONTAINER ONLY).  Same game plan / digest convention as gen_golden.py, with the full observation included:
digest = sha256 over players ascending of mask(u8) + partial_observation + full_observation bytes, first 8 bytes.
Output: tests/golden/games_both_<variant>.npz and one fully expanded micro game (expanded_both_micro.npz).
"""
import hashlib
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


def digest_both(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(obs[p]['valid_actions_mask'].astype(np.uint8).tobytes())
        h.update(obs[p]['partial_observation'].tobytes())
        h.update(obs[p]['full_observation'].tobytes())
    return int.from_bytes(h.digest()[:8], 'little')


def generate(channel_mode='extended', prefix='games_both', plan=(('barrage', 16), ('standard', 2), ('tiny', 32), ('micro', 32),
                                                                 ('fives', 16)), seed_offset=5000):
    ref = import_reference()
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    VC = ref.maenv.VERSION_CONFIGS
    norm = {}
    for name, n_games in plan:
        cfg = VC[GV(name)]
        R, C, U = cfg['rows'], cfg['columns'], cfg['initial_state_usable_rows']
        table = S.load_setup_table(G.HUMAN[name]) if name in G.HUMAN else None
        rs = random.Random(1000 + len(name))
        recs = []
        for gi in range(n_games):
            seed = G.BASE_SEED + seed_offset + gi
            if table is not None:
                i1 = orc.rng_below(orc.rng(seed, 0, 0, 0, 0), table.shape[0])
                i2 = orc.rng_below(orc.rng(seed, 0, 0, 0, 1), table.shape[0])
                m1, m2 = S.own_side_maps(table[i1], table[i2], R, C, U)
            else:
                m1, m2 = G.own_side_random_maps(cfg, rs)
            env = ref.maenv.StrategoMultiAgentEnv({'version': GV(name), 'observation_mode': OM.BOTH_OBSERVATIONS,
                                                   'obs_channel_mode': channel_mode})
            norm[name] = {k: [float(x) for x in getattr(env, '_' + k).reshape(-1)]
                          for k in ('p_obs_mids', 'p_obs_ranges', 'f_obs_mids', 'f_obs_ranges')}
            ob = np.zeros((R, C), dtype=np.int64)
            for loc in cfg['obstacle_locations']:
                ob[loc] = 1
            obs = env.reset(initial_state_override=env.base_env.create_initial_state(ob, m1, m2, cfg['max_turns']))
            rec = dict(m1=m1, m2=m2, actions=[], digests=[], init_digest=digest_both(obs), dones=[])
            if name == 'micro' and gi == 0 and channel_mode == 'extended':
                rec['full'] = [obs[1]['full_observation']]
            while True:
                p = list(obs.keys())[0]
                a = orc.sample_action(obs[p]['valid_actions_mask'].astype(np.uint8), seed, 0, 0, int(env.state[5, 0, 0]))
                obs, rew, done, info = env.step({p: a})
                rec['actions'].append(a)
                rec['digests'].append(digest_both(obs))
                rec['dones'].append(bool(done['__all__']))
                if 'full' in rec:
                    for pl in sorted(obs.keys(), reverse=True):
                        rec['full'].append(obs[pl]['full_observation'])
                if done['__all__']:
                    break
            recs.append(rec)
        off = np.cumsum([0] + [len(r['actions']) for r in recs]).astype(np.int64)
        out = dict(offsets=off, p1_maps=np.asarray([r['m1'] for r in recs], dtype=np.int8),
                   p2_maps=np.asarray([r['m2'] for r in recs], dtype=np.int8),
                   actions=np.concatenate([np.asarray(r['actions'], dtype=np.int32) for r in recs]),
                   digests=np.concatenate([np.asarray(r['digests'], dtype=np.uint64) for r in recs]),
                   dones=np.concatenate([np.asarray(r['dones'], dtype=np.uint8) for r in recs]),
                   init_digests=np.asarray([r['init_digest'] for r in recs], dtype=np.uint64))
        np.savez_compressed(os.path.join(G.GOLD, '%s_%s.npz' % (prefix, name)), **out)
        if name == 'micro' and channel_mode == 'extended':
            np.savez_compressed(os.path.join(G.GOLD, 'expanded_both_micro.npz'), p1_map=recs[0]['m1'].astype(np.int8),
                                p2_map=recs[0]['m2'].astype(np.int8), actions=np.asarray(recs[0]['actions'], dtype=np.int32),
                                full=np.asarray(recs[0]['full'], dtype=np.float32))
        print(prefix, name, n_games, 'games', int(off[-1]), 'steps')
    return norm


def main():
    generate()


if __name__ == '__main__':
    main()
