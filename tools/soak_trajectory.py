"""Soak of the per-step oracle pinning of the MULTI-STEP kernels (tests/test_gpu_trajectory.py): rollouts into trajectory buffers over fresh
seeds, ragged batch sizes, chunk lengths and garbage rates -- every slot (mask, observation(s), rewards, flags, player, drawn action) against
the oracle stepped alongside -- for a wall-clock budget.      python tools/soak_trajectory.py [seconds=300]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_trajectory as T  # noqa: E402

# (name, envs, chunk, calls, garbage, both, emit_obs)
PLAN = [('barrage', 41, 64, 8, 0.15, False, True), ('micro', 150, 30, 6, 0.2, False, True), ('standard', 11, 56, 6, 0.05, False, True),
        ('tiny', 90, 48, 4, 0.2, False, True), ('fives', 37, 40, 5, 0.15, False, True), ('octa_barrage', 27, 64, 5, 0.1, False, True),
        ('medium', 35, 50, 5, 0.1, False, True), ('short_barrage', 50, 50, 5, 0.1, False, True), ('barrage', 23, 40, 5, 0.1, True, True),
        ('barrage', 45, 64, 6, 0.2, False, False), ('standard', 13, 40, 5, 0.1, False, False), ('octa_barrage', 31, 50, 4, 0.2, False, False),
        ('medium', 43, 40, 4, 0.2, False, False), ('standard2', 4, 24, 3, 0.05, False, True),
        # whole workgroups only (multiples of 16 games, kept as they are): the launches that take the barrier per step beyond 8 slots (sgx_set_steps_barrier)
        ('barrage', 48, 64, 6, 0.15, False, True), ('standard', 16, 48, 4, 0.05, False, True), ('octa_barrage', 32, 56, 4, 0.1, False, True),
        ('medium', 48, 40, 4, 0.1, False, True), ('fives', 32, 40, 4, 0.15, False, True), ('barrage', 32, 30, 4, 0.1, True, True), ('standard2', 16, 20, 2, 0.05, False, True)]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    t0, salt, runs, steps = time.time(), 1, 0, 0
    while time.time() - t0 < budget:
        for name, n, chunk, calls, g, both, emit_obs in PLAN:
            n_envs = n if n % 16 == 0 else n + salt % 5
            T.test_every_step_of_a_multi_step_launch_equals_the_oracle(name, n_envs, chunk + salt % 7, calls, g, both=both, emit_obs=emit_obs, seed_salt=salt)
            runs += 1
            steps += n_envs * (chunk + salt % 7) * calls
            if time.time() - t0 > budget:
                break
        salt += 1
    print("trajectory soak ok: %d runs, %d env steps of multi-step launches compared slot by slot against the oracle in %.0f s" % (runs, steps, time.time() - t0))


if __name__ == '__main__':
    main()
