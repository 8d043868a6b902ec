"""In-process A/B of two builds of the library (kernel experiments: -DSGX_WPB, occupancy promises, ...) on THE SAME output buffers:
one env per build, same seed, the second env's output tensors and ring sets are the first one's -- so the allocation lottery of
plain buffers cannot decide the comparison.  Interleaved rounds; us per step in place and into a ring of
three sets, as multi-step launches and as one launch per step; with --tune the buffers come from the placement search.

    SGX_ALLOW_FOREIGN_BUILD=1 python tools/lib_ab.py barrage 65536 tools/_dev/a.so tools/_dev/b.so [--tune] [--steps 128] [--rounds 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    fn(steps)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('version')
    ap.add_argument('games', type=int)
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--tune', action='store_true')
    args = ap.parse_args()
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    n_libs = len(args.libs)
    envs = [VecStrategoEnv(args.version, args.games, seed=77, auto_reset=True, lib_path=p) for p in args.libs]
    first = envs[0]
    if args.tune:
        first.tune_placement(wide_extra_bytes=64 << 30)
    for e in envs:
        e.reset()
        e.rollout_steps(37)
    same = all(torch.equal(first.obs, e.obs) and torch.equal(first.mask, e.mask) and torch.equal(first.env_info(), e.env_info()) for e in envs[1:])
    first.alloc_output_ring(3, tune=args.tune, wide_extra_bytes=(64 << 30) if args.tune else 0)
    for e in envs[1:]:                       # the other builds write where the first one does
        e.obs, e.mask, e.fobs = first.obs, first.mask, first.fobs
        e._ring = first._ring
        e._ring_owners = first._ring_owners
        e._ring_pos = 0
        e._ring_ios = (_lib.SgxStepIO * len(first._ring))()
    res = {}
    for rnd in range(args.rounds):
        for i, e in enumerate(envs):
            for multi in (True, False):
                e.set_multi_step(multi)
                for what, fn in (('in place', lambda k: e.rollout_steps(k)), ('ring of 3', lambda k: e.rollout_steps(k, ring=True)),
                                 ('mask only', lambda k: e.rollout_steps(k, emit_obs=False)),
                                 ('no outputs', lambda k: e.rollout_steps(k, emit_obs=False, emit_mask=False))):
                    fn(8)
                    res.setdefault((what, multi, i), []).append(timed(fn, args.steps))
    for e in envs:
        e.close()
    del envs, first
    torch.cuda.empty_cache()
    try:                                                      # compact outputs: the second build writes the first one's tensors as well
        cenvs = [VecStrategoEnv(args.version, args.games, seed=77, auto_reset=True, compact_outputs=True, lib_path=p) for p in args.libs]
        for e in cenvs:
            e.reset()
            e.rollout_steps(16)
        for e in cenvs[1:]:
            e.obs, e.mask = cenvs[0].obs, cenvs[0].mask
        for rnd in range(args.rounds):
            for i, e in enumerate(cenvs):
                for multi in (True, False):
                    e.set_multi_step(multi)
                    e.rollout_steps(8)
                    res.setdefault(('compact', multi, i), []).append(timed(lambda k: e.rollout_steps(k), args.steps))
        for e in cenvs:
            e.close()
    except Exception as ex:      # noqa: BLE001
        print("  (no compact outputs: %s)" % ex)
    print("%s %d games, %s buffers, same results from every build: %s" % (args.version, args.games, 'tuned' if args.tune else 'plain', same))
    for what in ('in place', 'ring of 3', 'compact', 'mask only', 'no outputs'):
        if (what, True, 0) not in res:
            continue
        for multi in (True, False):
            print("   %-10s %-10s %s" % (what, 'multi-step' if multi else 'per-step',
                                         '   '.join('%s %7.2f us (%s)' % (os.path.basename(args.libs[i]), min(res[(what, multi, i)]),
                                                                          ' '.join('%.1f' % v for v in res[(what, multi, i)]))
                                                    for i in range(n_libs))), flush=True)


if __name__ == '__main__':
    main()
