"""Soak of the functional operator API against the oracle: random reachable states (exported from rollouts of random length),
random 1-D actions (valid and invalid, with and without the oscillation flag) -> next state / validity / masks in both
encodings / all four observation kinds, compared with OracleRules for a wall-clock budget.

    python tools/soak_procedural.py [seconds=180]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from stratego_env_amd.config import VARIANTS  # noqa: E402
from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

PLAN = [('barrage', 96), ('tiny', 128), ('micro', 128), ('fives', 96), ('standard', 24), ('octa_barrage', 64), ('medium', 64)]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
    t0, rounds, checked = time.time(), 0, 0
    envs = {name: (VecStrategoEnv(name, n, seed=1234, auto_reset=True), BatchedStrategoProceduralEnv(name, n)) for name, n in PLAN}
    for env, _ in envs.values():
        env.reset()
    rng = np.random.RandomState(99)
    pools = {}
    while time.time() - t0 < budget:
        for name, n in PLAN:
            env, penv = envs[name]
            v = VARIANTS[name]
            ru = orc.OracleRules(v.rows, v.columns)
            env.rollout_steps(int(rng.randint(1, 40)))
            states_t, players_t = env.export_state()
            states, players = states_t.cpu().numpy(), players_t.cpu().numpy()
            m1 = penv.get_valid_moves_as_1d_mask(states_t, players_t).cpu().numpy()
            ms = penv.get_valid_moves_as_spatial_mask(states_t, players_t).cpu().numpy()
            osc = bool(rng.randint(2))
            acts = np.asarray([int(rng.choice(np.flatnonzero(m1[i]))) if rng.rand() < 0.7 else int(rng.randint(ru.action_size))
                               for i in range(n)], dtype=np.int64)
            ns, npl, ok = penv.get_next_state(states_t, players_t, acts, allow_piece_oscillation=osc)
            ns, npl, ok = ns.cpu().numpy(), npl.cpu().numpy(), ok.cpu().numpy()
            # the same transitions pool to pool on packed records (sgx_expand: the no-observation kernel kind), successors' 1-D masks
            # in the same launch, through a shuffled parent index
            perm = rng.permutation(n).astype(np.int32)
            nodes = penv.pack(states_t, players_t)
            kids = pools.setdefault(name, penv.new_packed())
            m1_kids = torch.empty((n, ru.action_size), dtype=torch.uint8, device=states_t.device)
            ok_p, pl_p = kids.expand(nodes, acts[perm], parent_index=torch.as_tensor(perm), allow_piece_oscillation=osc, mask_1d_out=m1_kids)
            ns_p, npl_p = kids.unpack()
            ns_p, npl_p, ok_p, m1_kids = ns_p.cpu().numpy(), npl_p.cpu().numpy(), ok_p.cpu().numpy(), m1_kids.cpu().numpy()
            assert np.array_equal(ns_p, ns[perm]) and np.array_equal(ok_p, ok[perm]) and np.array_equal(npl_p, pl_p.cpu().numpy()), (name, 'expand')
            assert np.array_equal(npl_p[ok_p], npl[perm][ok_p]), (name, 'expand players')
            for i in rng.choice(n, size=min(n, 12), replace=False):
                assert np.array_equal(m1_kids[i], ru.get_valid_moves_as_1d_mask(ns_p[i], int(npl_p[i]))), (name, i, 'expand mask')
            nodes.close()
            obs = [fn(states_t, players_t).cpu().numpy() for fn in (
                penv.get_partially_observable_observation_extended_channels, penv.get_fully_observable_observation_extended_channels,
                penv.get_partially_observable_observation, penv.get_fully_observable_observation)]
            for i in range(n):
                st, pl = states[i], int(players[i])
                assert np.array_equal(m1[i], ru.get_valid_moves_as_1d_mask(st, pl)), (name, i)
                assert np.array_equal(ms[i], ru.get_valid_moves_as_spatial_mask(st, pl)), (name, i)
                try:
                    want, wpl = ru.get_next_state(st, pl, int(acts[i]), allow_piece_oscillation=osc)
                    assert ok[i] and np.array_equal(ns[i], want) and npl[i] == wpl, (name, i, int(acts[i]))
                except ValueError:
                    assert not ok[i] and np.array_equal(ns[i], st), (name, i, int(acts[i]))
                for o, fn in zip(obs, (ru.get_partially_observable_observation_extended_channels,
                                       ru.get_fully_observable_observation_extended_channels,
                                       ru.get_partially_observable_observation, ru.get_fully_observable_observation)):
                    assert o[i].tobytes() == fn(st, pl).tobytes(), (name, i, fn.__name__)
            checked += n
        rounds += 1
    print("procedural soak ok: %d rounds, %d states checked against the oracle in %.0f s" % (rounds, checked, time.time() - t0))


if __name__ == '__main__':
    main()
