"""Generate tests/golden/* from the imported REFERENCE (BUILD CONTAINER ONLY).

Everything written here is produced by running /root/reference (stub-imported, see ref_stubs.py):
inputs (own-side piece maps, flat spatial actions) and the reference's outputs for them.  The files are
data -- no reference source text.  Regenerate with:   python -m tools.oracle.gen_golden

Files:
  variants.json            the reference's VERSION_CONFIGS (rows, columns, max_turns, obstacles, piece
                           amounts, usable rows) + derived sizes + obs normalisation mids/ranges + DEFAULT_CONFIG keys
  index_tables.npz         per variant: spatial flat index -> 1-D index (penv:130-133) and the player -1
                           perspective map of every 1-D index (penv:110-115)
  games_<variant>.npz      N games: setups, action lists (valid + injected garbage), per-step digests of the
                           returned obs dicts, rewards, dones, errors, final int64 state (stored int16)
  expanded_<variant>.npz   a few games with every step's mask/obs stored in full
  kat.json                 known answers: SURVEY 8c rolling hashes, two-square sequence, RNG KAT inputs

Digest of one returned obs dict: sha256 over, for each player key in ascending order (-1 before 1),
mask.astype(uint8).tobytes() + partial_observation.tobytes(); first 8 bytes little-endian uint64.

Barrage / standard games use the bench's synthetic-rollout rule: setups
(i1, i2) = so_rng(seed, g, j, SETUP, 0/1) scaled into the Gravon table, action at turn t = the k-th valid
action with k = so_rng(seed, g, j, ACTION, t) scaled into nvalid.  The rule only CHOOSES inputs; the
recorded outputs are the reference's.  Held-out seeds: BASE_SEED + 1 .. BASE_SEED + n, env id 0.
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
from oracle import oracle as orc  # noqa: E402  (only its RNG / k-th-valid rule is used to choose inputs)
from stratego_env_amd import setups as S  # noqa: E402  (packed tables; verified against the reference by pack_inits)

GOLD = os.path.join(ROOT, 'tests', 'golden')
BASE_SEED = 0x5712A7E60
HUMAN = {'standard': 'standard', 'medium_standard': 'standard', 'short_standard': 'standard',
         'barrage': 'barrage', 'short_barrage': 'barrage'}


def digest_obs(obs):
    h = hashlib.sha256()
    for p in sorted(obs.keys()):
        h.update(obs[p]['valid_actions_mask'].astype(np.uint8).tobytes())
        h.update(obs[p]['partial_observation'].tobytes())
    return int.from_bytes(h.digest()[:8], 'little')


def make_env(ref, name):
    GV, OM = ref.enums.GameVersions, ref.enums.ObservationModes
    return ref.maenv.StrategoMultiAgentEnv({'version': GV(name), 'observation_mode': OM.PARTIALLY_OBSERVABLE})


def ref_step(env, p, a):
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            return env.step({p: a}), False
    except (ValueError, AssertionError):
        return None, True


def own_side_random_maps(cfg, rs):
    """Explicit toy setups (the reference would use random.shuffle, util:13-30; goldens pin explicit maps)."""
    R, C, U = cfg['rows'], cfg['columns'], cfg['initial_state_usable_rows']
    maps = []
    for _ in range(2):
        m = np.zeros((R, C), dtype=np.int64)
        cells = [(r, c) for r in range(U) for c in range(C)]
        rs.shuffle(cells)
        k = 0
        for pt, n in cfg['piece_amounts'].items():
            for _ in range(n):
                m[cells[k]] = pt.value
                k += 1
        maps.append(m)
    return maps


def play(ref, name, cfg, m1, m2, choose, garbage_rate, rs, expand=False, max_steps=None):
    env = make_env(ref, name)
    R, C = cfg['rows'], cfg['columns']
    ob = np.zeros((R, C), dtype=np.int64)
    for loc in cfg['obstacle_locations']:
        ob[loc] = 1
    init = env.base_env.create_initial_state(ob, m1, m2, cfg['max_turns'])
    obs = env.reset(initial_state_override=init)
    K = int(env.base_env.spatial_action_size[2])
    NA = R * C * K
    rec = dict(actions=[], errors=[], digests=[], rew=[], done=[], player=[], init_digest=digest_obs(obs))
    if expand:
        rec['masks'] = [obs[1]['valid_actions_mask'].astype(np.uint8)]
        rec['obs'] = [obs[1]['partial_observation']]
        rec['slot_player'] = [1]
    n = 0
    while True:
        p = list(obs.keys())[0]
        mask = obs[p]['valid_actions_mask'].reshape(-1)
        if garbage_rate and rs.random() < garbage_rate:
            kind = rs.randrange(4)
            a = [rs.randrange(NA), rs.randrange(R * C) * K + K - 1, rs.choice([-1, NA, NA + 5]),
                 (R * C - 1) * K + rs.randrange(K)][kind]
        else:
            a = choose(mask, n, int(env.state[5, 0, 0]))
        out, err = ref_step(env, p, a)
        rec['actions'].append(a)
        rec['errors'].append(err)
        if err:
            rec['digests'].append(0); rec['rew'].append((0.0, 0.0)); rec['done'].append(False); rec['player'].append(env.player)
            continue
        n += 1
        obs, rew, done, info = out
        rec['digests'].append(digest_obs(obs))
        rec['done'].append(bool(done['__all__']))
        rec['player'].append(int(env.player))
        rec['rew'].append((float(rew.get(1, 0.0)), float(rew.get(-1, 0.0))))
        if expand:
            for pl in sorted(obs.keys(), reverse=True):   # +1 first, then -1 (terminal steps have both)
                rec['masks'].append(obs[pl]['valid_actions_mask'].astype(np.uint8))
                rec['obs'].append(obs[pl]['partial_observation'])
                rec['slot_player'].append(pl)
        if done['__all__'] or (max_steps and n >= max_steps):
            rec['final_state'] = env.state.copy()
            rec['ending_invalid'] = bool(done['__all__'] and info[1]['game_result_was_invalid'])
            rec['finished'] = bool(done['__all__'])
            return rec


def pack_games(recs, m1s, m2s, extra=None):
    off = np.cumsum([0] + [len(r['actions']) for r in recs]).astype(np.int64)
    d = dict(
        offsets=off,
        p1_maps=np.asarray(m1s, dtype=np.int8), p2_maps=np.asarray(m2s, dtype=np.int8),
        actions=np.concatenate([np.asarray(r['actions'], dtype=np.int64) for r in recs]).astype(np.int32),
        errors=np.concatenate([np.asarray(r['errors'], dtype=np.uint8) for r in recs]),
        digests=np.concatenate([np.asarray(r['digests'], dtype=np.uint64) for r in recs]),
        rewards=np.concatenate([np.asarray(r['rew'], dtype=np.float32).reshape(-1, 2) for r in recs]),
        dones=np.concatenate([np.asarray(r['done'], dtype=np.uint8) for r in recs]),
        players=np.concatenate([np.asarray(r['player'], dtype=np.int8) for r in recs]),
        init_digests=np.asarray([r['init_digest'] for r in recs], dtype=np.uint64),
        final_states=np.asarray([r['final_state'] for r in recs], dtype=np.int16),
        ending_invalid=np.asarray([r['ending_invalid'] for r in recs], dtype=np.uint8),
        finished=np.asarray([r['finished'] for r in recs], dtype=np.uint8),
    )
    if extra:
        d.update(extra)
    return d


def main():
    ref = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    VC = ref.maenv.VERSION_CONFIGS

    # ---- variants.json -------------------------------------------------------------------------
    vj = {}
    for gv, cfg in VC.items():
        env = make_env(ref, gv.value)
        pe = env.base_env
        vj[gv.value] = dict(
            rows=cfg['rows'], columns=cfg['columns'], max_turns=cfg['max_turns'],
            obstacle_locations=[list(x) for x in cfg['obstacle_locations']],
            piece_counts=[cfg['piece_amounts'][ref.impl.SP(t)] for t in range(1, 13)],
            initial_state_usable_rows=cfg['initial_state_usable_rows'],
            action_size=int(pe.action_size), spatial_action_size=[int(x) for x in pe.spatial_action_size],
            discrete_n=int(env.action_space.n),
            p_obs_mids=[float(x) for x in env._p_obs_mids.reshape(-1)],
            p_obs_ranges=[float(x) for x in env._p_obs_ranges.reshape(-1)],
            f_obs_mids=[float(x) for x in env._f_obs_mids.reshape(-1)],
            f_obs_ranges=[float(x) for x in env._f_obs_ranges.reshape(-1)],
            human_inits_supported=gv.value in HUMAN,
        )
    dc = {k: (v.value if hasattr(v, 'value') else v) for k, v in ref.maenv.DEFAULT_CONFIG.items()}
    enums = {e.__name__: {m.name: m.value for m in e} for e in
             (ref.enums.ObservationModes, ref.enums.ObservationComponents, ref.enums.GameVersions)}
    json.dump(dict(variants=vj, default_config=dc, enums=enums), open(os.path.join(GOLD, 'variants.json'), 'w'), indent=1)

    # ---- index tables ---------------------------------------------------------------------------
    it = {}
    for name in ('barrage', 'octa_barrage', 'medium', 'fives', 'tiny', 'micro'):
        cfg = VC[ref.enums.GameVersions(name)]
        R, C = cfg['rows'], cfg['columns']
        pe = ref.penv.StrategoProceduralEnv(R, C)
        K = int(pe.spatial_action_size[2])
        s2o = np.zeros(R * C * K, dtype=np.int32)
        for a in range(R * C * K):
            s2o[a] = pe.get_action_1d_index_from_spatial_index(np.unravel_index(a, (R, C, K)))
        flip = np.asarray([pe.get_action_1d_index_from_player_perspective(i, -1) for i in range(int(pe.action_size))],
                          dtype=np.int32)
        it[name + '_spatial_to_1d'] = s2o
        it[name + '_flip_1d'] = flip
    np.savez_compressed(os.path.join(GOLD, 'index_tables.npz'), **it)

    # ---- games ----------------------------------------------------------------------------------
    plan = [  # name, n_games, garbage_rate, expanded games
        ('barrage', 256, 0.0, 1), ('standard', 24, 0.0, 0), ('short_barrage', 32, 0.1, 0), ('short_standard', 8, 0.1, 0),
        ('octa_barrage', 48, 0.1, 1), ('medium', 64, 0.1, 1), ('fives', 64, 0.1, 1), ('tiny', 128, 0.1, 2),
        ('micro', 128, 0.1, 2), ('standard2', 2, 0.05, 0),
    ]
    kat = {}
    for name, n_games, garbage, n_expand in plan:
        cfg = VC[ref.enums.GameVersions(name)]
        R, C, U = cfg['rows'], cfg['columns'], cfg['initial_state_usable_rows']
        counts = [cfg['piece_amounts'][ref.impl.SP(t)] for t in range(1, 13)]
        rs = random.Random(hash(name) & 0xFFFF)
        rs = random.Random(sum(ord(ch) for ch in name))
        recs, m1s, m2s, extra = [], [], [], {}
        table = S.load_setup_table(HUMAN[name]) if name in HUMAN else None
        cv = orc.make_cvariant(R, C, cfg['max_turns'], cfg['obstacle_locations'], counts, U, setups=table)
        seeds = []
        for gi in range(n_games):
            seed = BASE_SEED + 1 + gi
            seeds.append(seed)
            if table is not None:
                i1 = orc.rng_below(orc.rng(seed, 0, 0, 0, 0), table.shape[0])
                i2 = orc.rng_below(orc.rng(seed, 0, 0, 0, 1), table.shape[0])
                m1, m2 = S.own_side_maps(table[i1], table[i2], R, C, U)
            else:
                m1, m2 = own_side_random_maps(cfg, rs)

            def choose(mask, n, turn, seed=seed):
                return orc.sample_action(mask.astype(np.uint8), seed, 0, 0, turn)

            rec = play(ref, name, cfg, m1, m2, choose, garbage, rs, max_steps=(600 if name == 'standard2' else None))
            recs.append(rec); m1s.append(m1); m2s.append(m2)
        extra['seeds'] = np.asarray(seeds, dtype=np.uint64)
        np.savez_compressed(os.path.join(GOLD, 'games_%s.npz' % name), **pack_games(recs, m1s, m2s, extra))
        print(name, n_games, 'games', sum(len(r['actions']) for r in recs), 'actions', flush=True)
        if n_expand:
            ex = {}
            for gi in range(n_expand):
                seed = BASE_SEED + 1 + gi

                def choose(mask, n, turn, seed=seed):
                    return orc.sample_action(mask.astype(np.uint8), seed, 0, 0, turn)

                rec = play(ref, name, cfg, m1s[gi], m2s[gi], choose, 0.0, rs, expand=True,
                           max_steps=(48 if R * C >= 64 else None))
                ex['g%d_p1_map' % gi] = np.asarray(m1s[gi], dtype=np.int8)
                ex['g%d_p2_map' % gi] = np.asarray(m2s[gi], dtype=np.int8)
                ex['g%d_actions' % gi] = np.asarray(rec['actions'], dtype=np.int32)
                ex['g%d_masks' % gi] = np.asarray(rec['masks'], dtype=np.uint8)
                ex['g%d_obs' % gi] = np.asarray(rec['obs'], dtype=np.float32)
                ex['g%d_slot_player' % gi] = np.asarray(rec['slot_player'], dtype=np.int8)
                ex['g%d_rewards' % gi] = np.asarray(rec['rew'], dtype=np.float32)
                ex['g%d_dones' % gi] = np.asarray(rec['done'], dtype=np.uint8)
                ex['g%d_final_state' % gi] = rec['final_state'].astype(np.int16)
            np.savez_compressed(os.path.join(GOLD, 'expanded_%s.npz' % name), **ex)

    # ---- known answers ----------------------------------------------------------------
    from stratego_env.game.inits.barrage_human_inits import BARRAGE_INITS
    from stratego_env.game.inits.standard_human_inits import STANDARD_INITS
    for name, lst in (('barrage', BARRAGE_INITS), ('standard', STANDARD_INITS)):
        cfg = VC[ref.enums.GameVersions(name)]
        env = make_env(ref, name)
        obs = env.reset(initial_state_override=ref.util.create_game_from_data(lst[0], lst[1], cfg))
        m0 = obs[1]['valid_actions_mask']
        h = hashlib.sha256()
        n = 0
        init_obs_sha = hashlib.sha256(obs[1]['partial_observation'].tobytes()).hexdigest()[:16]
        init_mask_sha = hashlib.sha256(m0.astype(np.uint8).tobytes()).hexdigest()[:16]
        while True:
            p = list(obs.keys())[0]
            m = obs[p]['valid_actions_mask']
            a = int(np.flatnonzero(m)[(n * 7919) % int(m.sum())])
            obs, rew, done, info = env.step({p: a})
            n += 1
            for pl in sorted(obs.keys()):
                h.update(obs[pl]['valid_actions_mask'].astype(np.uint8).tobytes())
                h.update(obs[pl]['partial_observation'].tobytes())
            if done['__all__']:
                break
        kat[name] = dict(setup1=lst[0], setup2=lst[1], init_valid=[int(x) for x in np.flatnonzero(m0)],
                         init_obs_sha=init_obs_sha, init_mask_sha=init_mask_sha, steps=n,
                         rewards=[float(rew[1]), float(rew[-1])], rolling_sha=h.hexdigest()[:16],
                         invalid=bool(info[1]['game_result_was_invalid']))
    # two-square sequence on an empty 4x4 board
    pe = ref.penv.StrategoProceduralEnv(4, 4)
    m1 = np.zeros((4, 4), dtype=np.int64); m2 = np.zeros((4, 4), dtype=np.int64)
    m1[0, 0] = 5; m1[0, 3] = 11
    m2[0, 0] = 5; m2[0, 3] = 11
    st = pe.create_initial_state(np.zeros((4, 4), dtype=np.int64), m1, m2, 100)
    seq = []
    pl = 1
    moves = [((0, 0), (1, 0)), ((3, 3), (2, 3)), ((1, 0), (0, 0)), ((2, 3), (3, 3)), ((0, 0), (1, 0)), ((3, 3), (2, 3))]
    for (s, e) in moves:
        st, pl = pe.get_next_state(st, pl, pe.get_action_1d_index_from_positions(*s, *e))
    with contextlib.redirect_stdout(io.StringIO()):
        fourth_valid = bool(pe.is_move_valid_by_position(st, 1, 1, 0, 0, 0))
    kat['two_square'] = dict(p1_recent=[[int(x) for x in row] for row in st[6]],
                             fourth_oscillation_valid=fourth_valid,
                             p1_mask_after=[int(x) for x in np.flatnonzero(pe.get_valid_moves_as_spatial_mask(st, 1))])
    kat['base_seed'] = BASE_SEED
    json.dump(kat, open(os.path.join(GOLD, 'kat.json'), 'w'), indent=1)
    print(json.dumps({k: (v if k != 'barrage' and k != 'standard' else {kk: vv for kk, vv in v.items() if kk != 'init_valid'})
                      for k, v in kat.items()}, indent=1))


if __name__ == '__main__':
    main()
