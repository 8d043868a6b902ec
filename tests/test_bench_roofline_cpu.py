"""bench.py's roofline object: every fraction is priced on B_min, the packed layout's own byte minimum per env step -- never on a
count that overstates what the kernel moves -- and the SURVEY 8d formula only appears as a labelled comparison."""
import pytest
import types

import bench
from stratego_env_amd.config import VARIANTS


def test_b_min_is_the_packed_layouts_own_minimum():
    v = VARIANTS['barrage']
    assert bench.record_bytes(v) == 512 and bench.record_bytes(VARIANTS['standard']) == 640 and bench.record_bytes(VARIANTS['micro']) == 128
    # record in + record out + action in + next action out + observation + mask + results
    assert bench.b_min(v) == 2 * 512 + 4 + 4 + 4 * 67 * 100 + 3700 + 12 == 31544
    assert bench.b_min(VARIANTS['standard']) == 31800 and bench.b_min(VARIANTS['micro']) == 3624
    assert bench.b_min(v, full_obs=True) == 31544 + 4 * 79 * 100
    assert bench.b_min(v, rec_bytes=640) == 31800
    for name, var in VARIANTS.items():                       # the SURVEY formula assumes 32 dense boards: always more
        assert bench.b_min(var) < bench.b_alg(var.rows, var.columns), name


def test_roofline_fields():
    v = VARIANTS['barrage']
    r = bench.roofline('barrage', v, 65536, 260e-6, build_id='no-such-build')
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s'
    assert r['bytes_per_launch'] == 31544 * 65536 and r['b_min_bytes_per_step'] == 31544
    assert abs(r['achieved'] - 31544 * 65536 / 260e-6 / 1e9) < 1e-6 and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-12
    assert r['kernel'] == 'step_kernel<10,10,0,false>' and 'B_min = 31544 B' in r['frac_basis']
    # the counter bytes come from a committed PMC pass: static, and the source says whether it is this binary's
    assert r['traffic'] and r['traffic_source'].startswith('static, measured on an EARLIER binary: profiles/traffic.json[barrage]')
    assert 0.95 < r['traffic_over_b_min'] < 1.05              # the counters see what B_min predicts
    # no field called frac* is built on the SURVEY formula; it is a labelled comparison
    assert 'frac_algorithmic' not in r and 'frac_dram_min' not in r
    assert r['survey_8d']['bytes_per_step'] == 33848 and r['survey_8d']['gbps_if_those_bytes_moved'] > r['achieved']
    assert r['frac_dram'] is None and r['frac_untuned'] is None
    # with the rotating-outputs leg and the first allocation's observe time
    r2 = bench.roofline('barrage', v, 65536, 260e-6, first_us=330.0, rotating=300e-6)
    assert abs(r2['frac_dram'] - 31544 * 65536 / 300e-6 / 1e9 / 8000.0) < 1e-12 and r2['frac_dram'] < r2['frac']
    assert abs(r2['frac_untuned'] - 31544 * 65536 / 330e-6 / 1e9 / 8000.0) < 1e-12
    # every fraction of a physically possible launch time stays <= 1: the fastest launch ever measured (258.3 us) reads 1.0004 with
    # B_min -- the in-place figure is the memory side including the Infinity Cache, which is why frac_dram exists
    assert bench.roofline('barrage', v, 65536, 275e-6)['frac'] < 1.0
    # counter bytes scale per game; a workload without a counter entry has traffic None
    s = bench.roofline('standard', VARIANTS['standard'], 262144, 1.2e-3)
    h = bench.roofline('standard', VARIANTS['standard'], 131072, 0.6e-3)
    assert abs(h['traffic'] * 2 - s['traffic']) <= 2 and 0.8 < s['frac'] < 0.9
    f = bench.roofline('fives', VARIANTS['fives'], 65536, 100e-6)
    assert f['traffic'] is None and f['traffic_over_b_min'] is None and f['frac'] > 0


def test_traffic_source_names_the_binary(monkeypatch, tmp_path):
    import json
    import os
    (tmp_path / 'profiles').mkdir()
    json.dump({'barrage': {'games_per_launch': 65536, 'hbm_bytes_per_launch': 2069081498, 'source': 'x.txt', 'build_id': 'abc'}},
              open(tmp_path / 'profiles' / 'traffic.json', 'w'))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    t, src = bench.measured_traffic('barrage', 65536, build_id='abc')
    assert t == 2069081498 and src.startswith('static, measured on this binary')
    t, src = bench.measured_traffic('barrage', 65536, build_id='def')
    assert src.startswith('static, measured on an EARLIER binary')
    assert bench.measured_traffic('nothing', 1) == (None, None)
    assert os.path.exists(tmp_path / 'profiles' / 'traffic.json')


def test_placement_budgets_are_capped_by_free_memory(monkeypatch):
    """Eight ranks that share one host (or one GPU in the hardware test) search on their own: neither pass may take more than the
    free-memory fraction divided by the ranks on the device."""
    import torch
    args = types.SimpleNamespace(devices=None, placement_free_fraction=0.4, placement_wide_gb=64.0)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda *a: (100 << 30, 288 << 30))
    assert bench.placement_budgets(args, 8 << 30) == (8 << 30, 40 << 30)          # wide pass: 64 GB asked, 40 % of 100 GiB free
    args.devices = '0,0,0,0,0,0,0,0'
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert bench.ranks_sharing_device(args) == 8
    assert bench.placement_budgets(args, 8 << 30) == (5 << 30, 5 << 30)           # 40 GiB / 8 ranks
    args.devices = '0,1,2,3,4,5,6,7'
    assert bench.ranks_sharing_device(args) == 1


def test_an_extra_leg_that_cannot_run_leaves_the_line_but_a_parity_failure_does_not():
    """bench.optional_leg: resource failures of the extra legs become {"failed": ...} (and show in the summary at the end of the line); the
    SystemExit verify_against_oracle raises on a mismatch passes through."""
    import bench

    def no_memory(*a, **k):
        raise RuntimeError("HIP out of memory")

    def mismatch(*a, **k):
        raise SystemExit("bench.py: env 3 differs from the CPU oracle")
    leg = bench.optional_leg(no_memory, None, None, None, 'standard', 262144)
    assert leg["failed"].startswith("RuntimeError: HIP out of memory") and "262144" in leg["workload"]
    with pytest.raises(SystemExit):
        bench.optional_leg(mismatch, None, None, None)
    out = {"value": 1.0, "roofline": {}, "config": {"other_workloads": [leg] * 4, "trajectory": leg, "compact_outputs": None, "facade_n1": None}}
    s = bench.line_summary(out)
    assert s["config3_standard_262144"] == {"failed": leg["failed"]} and s["trajectory"] == {"failed": leg["failed"]}
