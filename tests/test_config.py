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


def test_custom_variants_and_geometry_libraries():
    """Boards the reference has no variant for: custom_variant() fills the usable rows, get_variant() takes the object, and the
    loader maps a size to the main library or to a library of its own (built on first use on a box with hipcc)."""
    import os
    import pytest
    from stratego_env_amd import build as hip_build
    from stratego_env_amd import _lib
    from stratego_env_amd.config import custom_variant, get_variant, VARIANTS
    v = custom_variant(7, 9)
    assert (v.rows, v.columns, v.initial_state_usable_rows) == (7, 9, 3) and v.piece_counts[10] == 1
    # one of every type, then scouts up to 8 like Standard: 19 of the 27 cells
    assert sum(v.piece_counts) == 19 and max(v.piece_counts) == v.piece_counts[1] == 8
    assert sum(custom_variant(5, 5).piece_counts) == 10
    # more than 8 pieces of one type are accepted (chained capture events, round 4); 128 are not
    assert custom_variant(10, 10, piece_counts=(0, 9, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0)).piece_counts[1] == 9
    with pytest.raises(ValueError):
        custom_variant(32, 32, piece_counts=(0, 128, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0))
    assert v.spatial_channels == 2 * 6 + 2 * 8 + 1 and v.action_size == 63 * 16 + 1 and get_variant(v) is v
    w = custom_variant(3, 3, max_turns=10, obstacle_locations=[(1, 1)], piece_counts=(0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0))
    assert w.obstacle_map()[1, 1] == 1 and w.pieces_per_side == 2 and w.max_turns == 10
    for bad in ((2, 5), (5, 2)):
        with pytest.raises(ValueError):
            custom_variant(*bad)
    with pytest.raises(ValueError):
        hip_build.build_geometry(33, 32)                      # more than 1024 cells
    assert {(x.rows, x.columns) for x in VARIANTS.values()} == set(hip_build.BUILTIN_GEOMETRIES)
    assert os.path.basename(hip_build.geometry_lib_path(7, 9)) == 'libstratego_mi355x_7x9.so'
    main = _lib.load()
    assert main.sgx_supports_geometry(10, 10) == 1 and main.sgx_supports_geometry(7, 9) == 0
    assert _lib.load_for_geometry(10, 10) is main and _lib.load_for_geometry(3, 4) is main
