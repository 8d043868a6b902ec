"""In-process A/B of non-temporal stores for the MASK (sgx_mask.h: emit_mask; the observation's policy is sgx_set_nt_stores): us per fused
rollout step in place and into a ring of three output sets, same env object and buffers, interleaved rounds.

    python tools/mask_nt_ab.py [--specs barrage:65536,...] [--steps 256] [--rounds 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402
from tools.lane_ab import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--specs', default='barrage:65536,standard:131072,octa_barrage:65536,micro:65536')
    args = ap.parse_args()
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    for spec in args.specs.split(','):
        name, n = spec.split(':')
        n = int(n)
        env = VecStrategoEnv(name, n, seed=0x5712A7E60, auto_reset=True)
        env.reset()
        env.tune_placement(max_extra_bytes=8 << 30) if env.obs.numel() * 4 > 300e6 else None
        env.rollout_steps(64)
        res = {}
        for rnd in range(args.rounds):
            for m in (0, 1):
                env._L.sgx_debug_set_mask_nt(env._h, m)
                env.rollout_steps(8)
                res.setdefault(('in place', m), []).append(timed(env.rollout_steps, args.steps))
        env.alloc_output_ring(3, tune=env.obs.numel() * 4 > 300e6)
        for rnd in range(args.rounds):
            for m in (0, 1):
                env._L.sgx_debug_set_mask_nt(env._h, m)
                env.rollout_steps(8, ring=True)
                res.setdefault(('ring of 3', m), []).append(timed(lambda k: env.rollout_steps(k, ring=True), args.steps))
        env._L.sgx_debug_set_mask_nt(env._h, -1)
        for what in ('in place', 'ring of 3'):
            a, b = min(res[(what, 0)]), min(res[(what, 1)])
            print("%-13s %7d games %-10s mask plain %8.2f us   mask non-temporal %8.2f us   nt / plain %.3f" % (name, n, what, a, b, b / a), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
