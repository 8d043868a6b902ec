"""Variant tables, enums and derived sizes vs the reference dump in tests/golden/variants.json."""
import numpy as np

from stratego_env_amd import enums
from stratego_env_amd.config import VARIANTS, get_variant
from tests.helpers import load_variants_json


def test_variants_match_reference_dump():
    ref = load_variants_json()['variants']
    assert sorted(ref) == sorted(VARIANTS)
    for name, r in ref.items():
        v = VARIANTS[name]
        assert (v.rows, v.columns, v.max_turns, v.initial_state_usable_rows) == \
            (r['rows'], r['columns'], r['max_turns'], r['initial_state_usable_rows']), name
        assert sorted(map(tuple, r['obstacle_locations'])) == sorted(v.obstacle_locations), name
        assert list(v.piece_counts) == r['piece_counts'], name
        assert v.action_size == r['action_size'], name
        assert list(v.spatial_action_size) == r['spatial_action_size'], name
        assert v.num_spatial_actions == r['discrete_n'], name
        assert bool(v.human_inits) == r['human_inits_supported'], name


def test_enums_match_reference_dump():
    ref = load_variants_json()['enums']
    for ename, members in ref.items():
        e = getattr(enums, ename)
        assert {m.name: m.value for m in e} == members


def test_get_variant_accepts_enum_and_string():
    assert get_variant(enums.GameVersions.BARRAGE) is VARIANTS['barrage']
    assert get_variant('micro').cells == 12


def test_captured_count_highs_quirk():
    # maenv:288-298: singleton / absent piece types keep hi = 8
    assert VARIANTS['barrage'].captured_count_highs() == (8, 2, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8)
    assert VARIANTS['standard'].captured_count_highs() == (8, 8, 5, 4, 4, 4, 3, 2, 8, 8, 8, 6)
    assert np.sum(VARIANTS['standard'].obstacle_map()) == 8
