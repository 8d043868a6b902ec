"""Logic-only rollouts (no observation pointer) on boards of at most 16 cells: the lane-per-game kernel, one launch per step (the default
since round 6) against the multi-step launch of the wave-per-game kernel (sgx_set_lane_kernel(h, 0)) -- advisor finding of round 5.
Same env sizes, interleaved rounds, microseconds per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd import _lib  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(env, k, **kw):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.rollout_steps(k, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k


for name in ('micro', 'tiny'):
    for n in (65536, 262144):
        a = VecStrategoEnv(name, n, seed=1, auto_reset=True); a.reset()
        b = VecStrategoEnv(name, n, seed=1, auto_reset=True); b.reset(); b.set_lane_kernel(False)
        for kw, what in (({'emit_obs': False}, 'mask only'), ({'emit_obs': False, 'emit_mask': False}, 'no outputs')):
            ta, tb = [], []
            for r in range(5):
                ta.append(timed(a, 64, **kw)); tb.append(timed(b, 64, **kw))
            ka, kb = a.last_launch_kind, b.last_launch_kind
            print("%-6s %7d games %-10s lane kernel per step (kind %d) %6.1f us | wave-per-game multi-step (kind %d) %6.1f us" %
                  (name, n, what, ka, min(ta[1:]), kb, min(tb[1:])), flush=True)
        a.close(); b.close()
