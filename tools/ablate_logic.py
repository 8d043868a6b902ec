"""What the parts of the game logic cost (10x10 boards): launch time of state-preserving observe launches WITHOUT any output (the
no-observation kind: staging + mask generation only) on a diagnostic build (-DSGX_ABLATE) whose handles skip parts of the logic by the
bits of SGX_MAP="0,<bits>" (sgx_mask.h: SGX_ABLATED), and of the logic-only step on the same build with nothing skipped.  The ablated
handles compute wrong masks by construction; they are only timed.  Every handle holds the same mid-game states (export / import).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -DSGX_BUILD_ID='"ablate"' -DSGX_ABLATE -DSGX_ONLY_EXTRA \
          -DSGX_EXTRA_R=10 -DSGX_EXTRA_C=10 -I include stratego_env_amd/csrc/stratego_mi355x.hip -o tools/_dev/ablate10.so
    SGX_ALLOW_FOREIGN_BUILD=1 python tools/ablate_logic.py [barrage|standard] [games]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SGX_ALLOW_FOREIGN_BUILD'] = '1'
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_dev', 'ablate10.so')
CASES = (('everything (staging + derived boards + mask generation)', 0), ('without the ray walk of pass 2', 1), ('pass 1 only (occupancy + compaction)', 16),
         ('without mask generation', 2), ('without mask generation and derived boards', 6), ('staging only', 8))


def make(version, n, bits):
    os.environ['SGX_MAP'] = '0,%d' % bits
    try:
        return VecStrategoEnv(version, n, seed=3, auto_reset=True, lib_path=LIB)
    finally:
        os.environ.pop('SGX_MAP', None)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    ref = make(version, n, 0)
    ref.reset()
    ref.rollout_steps(40)
    st, pl = ref.export_state()
    print("%s, %d games, observe launches without any output (us per launch):" % (version, n))
    base = None
    for name, bits in CASES:
        env = make(version, n, bits)
        env.import_state(st, pl)
        t = timed(lambda: env.observe(emit_obs=False, emit_mask=False))
        base = t if base is None else base
        print("  %-62s %7.1f us   (%+.1f)" % (name, t, t - base), flush=True)
        env.close()
    t_step = timed(lambda: ref.step(ref.next_actions, want_next_actions=True, emit_obs=False, emit_mask=False), reps=100)
    ref.sample_valid_actions()
    t_nosamp = timed(lambda: ref.step(ref.next_actions, want_next_actions=False, emit_obs=False, emit_mask=False), reps=100)
    print("  logic-only step (apply + mask generation + sampler + write-back)  %7.1f us;  without the sampler (mostly invalid actions) %7.1f us" % (t_step, t_nosamp))
    ref.close()


if __name__ == '__main__':
    main()
