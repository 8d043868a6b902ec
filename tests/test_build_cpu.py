"""stratego_env_amd/build.py on a box without the ROCm compiler: a shipped per-geometry library is used as it is, a size that was not
prebuilt fails with the instruction how to prebuild it."""
import os
import warnings

import pytest

from stratego_env_amd import build as B


def test_geometry_library_without_hipcc(monkeypatch, tmp_path):
    monkeypatch.setattr(B.shutil, 'which', lambda name: str(tmp_path / 'no-such-hipcc'))
    monkeypatch.setattr(B, 'OUT_DIR', str(tmp_path))
    shipped = B.geometry_lib_path(7, 7)
    with open(shipped, 'wb') as f:
        f.write(b'\x7fELF')
    os.utime(shipped, (1, 1))                              # far older than the sources
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert B.build_geometry(7, 7) == shipped
    assert any('hipcc is not available' in str(x.message) for x in w)
    with pytest.raises(RuntimeError, match=r'python -m stratego_env_amd.build 11x9'):
        B.build_geometry(11, 9)
    with pytest.raises(ValueError):
        B.build_geometry(2, 9)
    with pytest.raises(ValueError):
        B.build_geometry(40, 40)
