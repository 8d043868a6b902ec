"""Multi-step launches into rings of 1 .. 8 output sets: us per step and TB/s on the per-step byte minimum.  With all steps of a call in one
launch a wave rewrites its game's slot of a set every n_sets steps of ITS OWN (tens of microseconds), so a small ring can live in the
Infinity Cache where one launch per step swept the whole set: how many sets does a DRAM-side figure need?
    python tools/ring_size_probe.py [barrage] [65536] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import b_min  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'barrage'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    env = VecStrategoEnv(version, n, seed=5, auto_reset=True)
    env.reset()
    env.rollout_steps(40)
    v = env.variant
    for sets in (1, 2, 3, 4, 6, 8, 12, 16):
        try:
            env.alloc_output_ring(sets)
        except Exception as e:      # noqa: BLE001
            print("ring of %d: %s" % (sets, e))
            break
        row = []
        for multi in (True, False):
            env.set_multi_step(multi)
            env.rollout_steps(16, ring=True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            env.rollout_steps(steps, ring=True)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / steps * 1e3
            fused = steps if (multi and env.last_launch_kind >= 2) else 1
            byts = b_min(v, rec_bytes=env.record_bytes, fused_steps=min(fused, 256)) * n
            row.append("%s %7.1f us = %5.2f TB/s (kind %d)" % ('multi-step' if multi else 'per-step  ', us, byts / us / 1e6, env.last_launch_kind))
        print("%s %d games, ring of %2d sets (%.1f GB):  %s   |   %s" % (version, n, sets, sets * (env.obs.numel() * 4 + env.mask.numel()) / 1e9, row[0], row[1]), flush=True)
    env.close()


if __name__ == '__main__':
    main()
