"""A few multi-step launches (lane_steps_kernel) of 65,536 Micro / Tiny games, in place and into a ring of three output sets, for
rocprofv3 (--kernel-trace --stats, or --pmc ...): one launch = `steps` steps.   python tools/kstep_probe.py [micro] [65536] [256]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def main():
    version = sys.argv[1] if len(sys.argv) > 1 else 'micro'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    env = VecStrategoEnv(version, n, seed=0x5712A7E60, auto_reset=True)
    env.reset()
    env.rollout_steps(8)
    for _ in range(3):
        env.rollout_steps(steps)                 # in place
    env.alloc_output_ring(3)
    torch.cuda.synchronize()
    for _ in range(3):
        env.rollout_steps(steps, ring=True)      # ring of 3 (the launches AFTER the three in-place ones in the trace)
    torch.cuda.synchronize()
    env.close()


if __name__ == '__main__':
    main()
