"""bench.py's roofline object: counter-based fraction, its DRAM-side lower bound, the SURVEY-formula fraction."""
import bench
from stratego_env_amd.config import VARIANTS


def test_roofline_fields():
    v = VARIANTS['barrage']
    r = bench.roofline('barrage', v, 65536, 260e-6)
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s'
    assert r['traffic'] and r['traffic_source'].startswith('profiles/traffic.json[barrage]')
    assert abs(r['frac'] - r['traffic'] / 260e-6 / 1e9 / 8000.0) < 1e-12
    # the mask of 65,536 Barrage games (242 MB) fits the 256 MiB Infinity Cache: the lower bound leaves it out
    mask = 65536 * v.num_spatial_actions
    assert abs(r['frac_dram_min'] - (r['traffic'] - mask) / 260e-6 / 1e9 / 8000.0) < 1e-12 and r['frac_dram_min'] < r['frac']
    assert r['algorithmic_bytes_per_launch'] == bench.b_alg(10, 10) * 65536 == 2218262528 and r['frac_algorithmic'] > r['frac']
    # 262,144 Standard games: the mask (970 MB) cannot stay in the cache, the two fractions coincide; counter bytes scale per game
    s = bench.roofline('standard', VARIANTS['standard'], 262144, 1.2e-3)
    assert s['frac_dram_min'] == s['frac'] and 0.8 < s['frac'] < 0.9
    h = bench.roofline('standard', VARIANTS['standard'], 131072, 0.6e-3)
    assert abs(h['traffic'] * 2 - s['traffic']) <= 2
    # a workload without a counter entry falls back to the algorithmic bytes and says so
    f = bench.roofline('fives', VARIANTS['fives'], 65536, 100e-6)
    assert f['traffic'] is None and f['frac'] == f['frac_algorithmic'] and 'algorithmic' in f['frac_basis'] and f['frac_dram_min'] is None
