"""Launch time of the step kernel with outputs switched off one by one (a NULL output pointer skips its phase): what the
observation rendering, the mask emission and the fused sampler cost on a board size.   python tools/phase_cost.py [version] [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'micro'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    env = VecStrategoEnv(version, n, seed=3, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    for _ in range(40):
        env.rollout_step()

    def timed(obs, mask, fused, steps=64):        # (`env` is looked up at call time: the compact env below is timed by the same code)
        def one():
            env.step(env.next_actions, want_next_actions=fused, emit_obs=obs, emit_mask=mask)
            if not fused:
                pass        # (the same actions are replayed: mostly invalid afterwards, which skips the apply phase -- see 'apply')
        for _ in range(8):
            env.rollout_step()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            one()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps * 1e3

    full = timed(True, True, True)
    print("%s, %d games: full step %.1f us" % (version, n, full))
    print("  without the observation        %.1f us  (observation: %.1f us)" % (timed(False, True, True), full - timed(False, True, True)))
    print("  without the mask bytes         %.1f us" % timed(True, False, True))
    print("  without observation and mask   %.1f us" % timed(False, False, True))
    env.close()
    try:
        env = VecStrategoEnv(version, n, seed=3, auto_reset=True, compact_outputs=True)
    except Exception as e:                      # boards / modes without compact outputs
        print("  (no compact outputs: %s)" % e)
        return
    env.reset()
    env.sample_valid_actions()
    for _ in range(40):
        env.rollout_step()
    print("  compact outputs: full step %.1f us   codes only %.1f us   mask bits only %.1f us   neither %.1f us"
          % (timed(True, True, True), timed(True, False, True), timed(False, True, True), timed(False, False, True)))
    env.close()


if __name__ == '__main__':
    main()
