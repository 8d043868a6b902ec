"""Launches WITHOUT an observation (search expansion, mask-only and logic-only rollouts) with two games per wave (Geo<R, C, 2>: the default
where a board allows it) against one game per wave (SGX_HALF_WAVE=0): same library, one env object per setting (the setting is read when a
handle is created), interleaved rounds, microseconds per step of all games; the results must agree.
    python tools/half_wave_ab.py [variant ...]          AB_MODES=0,1,2 adds a development build's four-games-per-wave variant"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402
from stratego_env_amd.procedural_env import BatchedStrategoProceduralEnv  # noqa: E402

MODES = [int(x) for x in os.environ.get('AB_MODES', '0,1').split(',')]
LABEL = {0: 'one game per wave', 1: 'two games per wave', 2: 'four games per wave'}


def timed(fn, k):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k


def make(name, n, mode, **kw):
    os.environ['SGX_HALF_WAVE'] = str(mode)               # (development builds take 2 = four games per wave through the variable only)
    e = VecStrategoEnv(name, n, seed=9, auto_reset=True, **kw)
    if mode in (0, 1):
        e.set_half_wave(bool(mode))
    e.reset()
    e.rollout_steps(30)
    return e


def line(name, n, what, ts, same, unit='games'):
    cols = "   ".join("%s %7.1f us" % (LABEL[m], t) for m, t in zip(MODES, ts))
    print("%-12s %6d %s  %-34s %s   (x%s)  same results: %s" % (name, n, unit, what, cols, " / ".join("%.2f" % (ts[0] / t) for t in ts[1:]), same), flush=True)


def main():
    names = sys.argv[1:] or ['barrage', 'standard']
    n, K = 65536, 64
    for name in names:
        envs = [make(name, n, m) for m in MODES]
        for what, kw, multi in (('mask only, multi-step', {'emit_obs': False}, True), ('no outputs, multi-step', {'emit_obs': False, 'emit_mask': False}, True),
                                ('mask only, one launch per step', {'emit_obs': False}, False), ('no outputs, one launch per step', {'emit_obs': False, 'emit_mask': False}, False)):
            ts = [[] for _ in envs]
            for e in envs:
                e.set_multi_step(multi)
            for _ in range(4):
                for i, e in enumerate(envs):
                    ts[i].append(timed(lambda: e.rollout_steps(K, **kw), K))
            same = all(torch.equal(envs[0].env_info(), e.env_info()) and torch.equal(envs[0].next_actions, e.next_actions) and torch.equal(envs[0].reward, e.reward)
                       and (not kw.get('emit_mask', True) or torch.equal(envs[0].mask, e.mask)) for e in envs[1:])
            line(name, n, what, [min(t[1:]) for t in ts], same)
        # search expansion on packed records (sgx_expand: get_next_state pool to pool)
        states, players = envs[0].export_state()
        res = []
        for m in MODES:
            os.environ['SGX_HALF_WAVE'] = str(m)
            pe = BatchedStrategoProceduralEnv(name, n)
            m1 = pe.get_valid_moves_as_1d_mask(states, players)
            acts = torch.argmax((m1 != 0).to(torch.int8), dim=1).to(torch.int32)
            nodes = pe.pack(states, players)
            kids = pe.new_packed()
            kids.expand(nodes, acts)
            t = [timed(lambda: [kids.expand(nodes, acts) for _ in range(20)], 20) for _ in range(4)]
            st, pl = kids.unpack()
            res.append((min(t[1:]), st, pl))
        same = all(torch.equal(res[0][1], r[1]) and torch.equal(res[0][2], r[2]) for r in res[1:])
        line(name, n, 'search expansion (sgx_expand)', [r[0] for r in res], same, unit='states')
        print("%-12s        M states/s: %s" % (name, " / ".join("%.0f" % (n / r[0]) for r in res)), flush=True)
        for e in envs:
            e.close()


if __name__ == '__main__':
    main()
