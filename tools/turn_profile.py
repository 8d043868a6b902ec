"""Launch time of the step kernel by turn number, all games in lockstep from reset(): full outputs, no mask, no outputs at all
(three envs with the same seed, sharing the tuned output buffers).  The first 40 turns of a Barrage batch are not alike."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    turns = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    envs = []
    for i in range(3):
        e = VecStrategoEnv(version, n, seed=0x5712A7E60, auto_reset=True)
        if envs:
            e.obs, e.mask = envs[0].obs, envs[0].mask
        e.reset()
        if not envs:
            e.tune_placement()
        e.sample_valid_actions()
        envs.append(e)
    kw = [dict(), dict(emit_mask=False), dict(emit_obs=False, emit_mask=False)]
    print("turn   full   no-mask   logic-only   valid moves per game")
    for t in range(turns):
        row = []
        for e, k in zip(envs, kw):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            e.step(e.next_actions, want_next_actions=True, **k)
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) * 1e3)
        moves = float(envs[0].mask.sum(dtype=torch.int64)) / n
        print("%4d  %6.1f  %6.1f  %6.1f   %5.1f" % (t + 1, row[0], row[1], row[2], moves), flush=True)


if __name__ == '__main__':
    main()
