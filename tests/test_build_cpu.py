"""stratego_env_amd/build.py and the binary <-> source tie: the library carries a hash of the sources it was compiled from
(sgx_build_id), build.py rebuilds by that hash instead of file times, _lib.load() refuses a library built from other sources, and
every exported entry point that touches the device does so under the device guard."""
import os
import re
import shutil
import warnings

import pytest

from stratego_env_amd import _lib
from stratego_env_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_SRC = os.path.join(ROOT, 'stratego_env_amd', 'csrc', 'stratego_mi355x.hip')


def _fake_lib(path, build_id):
    with open(path, 'wb') as f:
        f.write(b'\x7fELF' + b'\0' * 32 + B.BUILD_ID_MARKER + build_id.encode() + b'\0' + b'\1' * 16)


def test_geometry_library_without_hipcc(monkeypatch, tmp_path):
    """A box without the ROCm compiler uses a shipped per-geometry library only if it was built from the sources that are there; a
    size that was not prebuilt fails with the instruction how to prebuild it."""
    monkeypatch.setattr(B.shutil, 'which', lambda name: str(tmp_path / 'no-such-hipcc'))
    monkeypatch.setattr(B, 'OUT_DIR', str(tmp_path))
    shipped = B.geometry_lib_path(7, 7)
    _fake_lib(shipped, B.source_hash())
    os.utime(shipped, (1, 1))                              # far older than the sources: file times do not matter any more
    assert B.build_geometry(7, 7) == shipped
    _fake_lib(shipped, '0123456789abcdef')                 # built from other sources, and no compiler to rebuild it with
    with pytest.raises(RuntimeError, match=r'built from other sources'):
        B.build_geometry(7, 7)
    with pytest.raises(RuntimeError, match=r'python -m stratego_env_amd.build 11x9'):
        B.build_geometry(11, 9)
    with pytest.raises(ValueError):
        B.build_geometry(2, 9)
    with pytest.raises(ValueError):
        B.build_geometry(40, 40)


def test_source_hash_covers_every_source(monkeypatch, tmp_path):
    files = B.source_files()
    names = {os.path.basename(f) for f in files}
    assert {'stratego_mi355x.hip', 'sgx_step.h', 'sgx_obs.h', 'sgx_mask.h', 'sgx_layout.h', 'sgx_setup.h', 'sgx_aux_kernels.h',
            'sgx_mem.h', 'stratego_mi355x.h'} <= names
    copies = []
    for f in files:
        c = tmp_path / os.path.basename(f)
        shutil.copy(f, c)
        copies.append(str(c))
    monkeypatch.setattr(B, 'source_files', lambda: copies)
    h0 = B.source_hash()
    assert re.fullmatch(r'[0-9a-f]{16}', h0)
    for c in copies:                                        # one more byte in ANY source file is another build
        with open(c, 'ab') as fh:
            fh.write(b'\n')
        h1 = B.source_hash()
        assert h1 != h0, c
        h0 = h1


def test_built_library_carries_the_hash_of_its_sources():
    path = B.build()
    assert B.read_build_id(path) == B.source_hash()
    assert B.is_current(path) and not B.needs_build()
    L = _lib.load()
    assert L.sgx_build_id().decode() == B.source_hash() == L.build_id


def test_prebuilt_geometry_libraries_are_current():
    """The libraries of board sizes outside the reference's variants (build_geometry) travel to the GPU box prebuilt; one left over from
    older sources would be rebuilt there on first use (half a minute of hipcc inside a GPU test).  __graft_entry__.build() refreshes
    them all: whatever is in the tree must carry the hash of the sources next to it."""
    import glob
    B.build()
    stale = [os.path.basename(f) for f in glob.glob(os.path.join(B.OUT_DIR, 'libstratego_mi355x_*x*.so')) if B.read_build_id(f) != B.source_hash()]
    assert not stale, "stale geometry libraries (run `python -c 'import __graft_entry__ as g; g.build()'`): %s" % stale


def test_stale_library_fails_loudly(tmp_path, monkeypatch):
    """The same binary under another build id -- what a library left over from older sources looks like -- must not load."""
    path = B.build()
    data = open(path, 'rb').read()
    at = data.index(B.BUILD_ID_MARKER) + len(B.BUILD_ID_MARKER)
    stale = tmp_path / 'libstratego_mi355x_stale.so'
    stale.write_bytes(data[:at] + b'0' * 16 + data[at + 16:])
    assert B.read_build_id(str(stale)) == '0' * 16 and not B.is_current(str(stale))
    monkeypatch.delenv('SGX_ALLOW_FOREIGN_BUILD', raising=False)
    with pytest.raises(_lib.SgxError, match='built from other sources'):
        _lib.load(str(stale))
    monkeypatch.setenv('SGX_ALLOW_FOREIGN_BUILD', '1')      # kernel experiments (tools/ A/B runs): loaded, loudly
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        L = _lib.load(str(stale))
    assert L.sgx_build_id() == b'0' * 16
    assert any('built from other sources' in str(x.message) for x in w)
    _lib._libs.pop(str(stale), None)


def test_every_entry_point_restores_the_callers_device():
    """Source scan: the only hipSetDevice calls are DeviceGuard's own, and every exported function that makes a HIP call or launches
    a kernel does it under SGX_ON_DEVICE / a DeviceGuard (a handle is bound to one device, different handles are
    independent -- one process may drive several GPUs)."""
    src = open(HIP_SRC).read()
    guard = src[src.index('struct DeviceGuard {'):src.index('#define SGX_ON_DEVICE')]
    rest = src.replace(guard, '')
    assert guard.count('hipSetDevice(') == 2 and 'hipSetDevice(' not in rest
    parts = re.split(r'\nSGX_API ', src)[1:]
    checked = 0
    for part in parts:
        head = part[:part.index('(')]
        name = head.split()[-1].lstrip('*')
        body = part[:part.index('\n}\n') + 3] if '\n}\n' in part else part
        if '{' not in body.split('\n', 1)[0] or body.split('\n', 1)[0].rstrip().endswith('}'):
            continue                                        # one-line getters touch no device
        touches = re.search(r'\bhip[A-Z]\w*\(|<<<|\blaunch_(step|import|export)\(|\bstep_single\(', body)
        if not touches:
            continue
        if name == 'sgx_destroy':
            assert 'DeviceGuard device_guard_(h->device);' in body
        else:
            assert 'SGX_ON_DEVICE(' in body, name
        checked += 1
    assert checked >= 20
