"""Soak run of the step-by-step GPU-vs-oracle parity check (tests/test_gpu_parity.py::test_step_bit_exact_vs_oracle) over
many seeds, ragged batch sizes and garbage-action rates, for a given wall-clock budget.

    python tools/soak_parity.py [seconds=300]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_parity as T  # noqa: E402
from tests import test_gpu_full_obs as F  # noqa: E402

PLAN = [('barrage', 37, 400, 0.1), ('micro', 131, 150, 0.2), ('tiny', 77, 200, 0.2), ('fives', 45, 200, 0.15),
        ('standard', 9, 350, 0.05), ('octa_barrage', 29, 300, 0.1), ('medium', 33, 250, 0.1), ('short_barrage', 21, 200, 0.1),
        ('short_standard', 6, 470, 0.05), ('medium_standard', 5, 300, 0.05), ('standard2', 3, 120, 0.05)]


CUSTOM_PLAN = [('c3x3', 41, 80, 0.2), ('c7x7', 27, 200, 0.1), ('c9x5', 25, 200, 0.1), ('c12x12', 9, 250, 0.1), ('c3x40', 9, 120, 0.1),
               ('c20x20', 4, 300, 0.1), ('c17x16', 5, 250, 0.1), ('c32x32', 2, 160, 0.1)]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    if len(sys.argv) > 2 and sys.argv[2] == 'custom':      # the custom geometries of tests/test_gpu_generic_geometry.py (boards of up to 1,024 cells)
        from stratego_env_amd import config
        from tests.test_gpu_generic_geometry import CUSTOM
        config.VARIANTS.update(CUSTOM)
        t0, salt, runs, steps = time.time(), 1, 0, 0
        while time.time() - t0 < budget:
            for name, n, t, g in CUSTOM_PLAN:
                T.test_step_bit_exact_vs_oracle(name, n + salt % 3, t, g, seed_salt=salt, require_endings=False)
                runs += 1
                steps += (n + salt % 3) * t
                if time.time() - t0 > budget:
                    break
            salt += 1
        print("custom-geometry soak ok: %d runs, %d env steps compared output by output against the oracle in %.0f s" % (runs, steps, time.time() - t0))
        return
    t0, salt, runs, steps = time.time(), 1, 0, 0
    while time.time() - t0 < budget:
        for name, n, t, g in PLAN:
            T.test_step_bit_exact_vs_oracle(name, n + salt % 5, t, g, seed_salt=salt)
            runs += 1
            steps += (n + salt % 5) * t
            if time.time() - t0 > budget:
                break
        F.check_both_obs_vs_oracle(('barrage', 'tiny', 'micro', 'fives')[salt % 4], 16 + salt % 7, 150, ('extended', 'original')[salt % 2])
        # the lane-per-game kernel (forced; it is the default only for launches without an observation) on the boards it plays
        for name, n, t, g in (('micro', 131, 150, 0.2), ('tiny', 77, 200, 0.2)):
            T.test_step_bit_exact_vs_oracle(name, n + salt % 64, t, g, seed_salt=salt, final_obs=False, lane_kernel=True)
            runs += 1
            steps += (n + salt % 64) * t
        # the no-observation kernel kind (steps without an observation pointer) on the boards with one game per wave
        for name, n, t, g in (('barrage', 37, 300, 0.1), ('octa_barrage', 29, 200, 0.1), ('standard', 9, 250, 0.05)):
            T.test_step_bit_exact_vs_oracle(name, n + salt % 5, t, g, seed_salt=salt, final_obs=False, emit_obs=False)
            runs += 1
            steps += (n + salt % 5) * t
        salt += 1
    print("soak ok: %d runs, %d env steps compared output by output against the oracle in %.0f s" % (runs, steps, time.time() - t0))


if __name__ == '__main__':
    main()
