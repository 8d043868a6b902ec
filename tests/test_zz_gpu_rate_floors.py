"""Rate floors (run LAST: a host that is busy or slow must not keep the parity suites from running under `pytest -x`)."""
import pytest

pytestmark = pytest.mark.gpu


def test_facade_n1_rate_floor():
    """BASELINE config 1 on the GPU: ONE Barrage game behind the dict API (loop:34-63), a random valid action per step.  The loop that
    bench.py reports as config.facade_n1 must stay above 30,000 env.step()/s (measured 35-40 k; the interpreted reference ~1.9 k, one
    thread of the CPU port ~30 k) -- a regression of the latency path (sgx_step_sync polling a host-mapped word, outputs written
    straight to pinned host memory) shows up here, not in the batched throughput figures.  The best of up to eight runs counts."""
    import sys
    import bench
    import bench_legs
    r = bench_legs.facade_leg(sys.modules['bench'], n_steps=2000, runs=8, good_enough=30000)     # (up to eight runs: a busy host core must not fail the suite)
    assert r['steps'] == 2000 and r['games_finished'] >= 1
    assert r['steps_per_s'] >= 30000, r
    assert r['env_step_calls_per_s'] >= r['steps_per_s']
