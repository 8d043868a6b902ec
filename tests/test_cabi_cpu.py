"""C-ABI library checks that need no GPU: it builds, loads, exports every declared symbol, its host-only
normalisation LUT is bit-exact against the reference dump, and it fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from stratego_env_amd import _lib
from stratego_env_amd import build as hip_build
from stratego_env_amd.config import VARIANTS
from tests.helpers import load_variants_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    hip_build.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'stratego_mi355x.h')).read()
    declared = sorted(set(re.findall(r'\b(sgx_[a-z0-9_]+)\s*\(', hdr)))
    assert declared == sorted(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.sgx_abi_version() == _lib.ABI_VERSION == 13


def test_struct_sizes_match_header():
    # sgx_config: 4 + 12 + 1 int32 + SGX_MAX_CELLS bytes; sgx_step_io: 12 pointers + 2 int32
    assert C.sizeof(_lib.SgxConfig) == 17 * 4 + 1024
    assert C.sizeof(_lib.SgxStepIO) == 12 * 8 + 8
    assert C.sizeof(_lib.SgxOutputs) == 3 * 8 + 4 * 8 + 2 * 4 + 2 * 64 * 4 + 2 * 4  # sgx_outputs of the header, field by field


def test_obs_lut_bit_exact_vs_reference_constants(lib):
    """LUT[ch][v] must equal (float32(v) - mid) / range computed the way numpy does (maenv:388-391, 506-508)."""
    ref = load_variants_json()['variants']
    for name, v in VARIANTS.items():
        cfg = _lib.make_config(v)
        lut = np.zeros(67 * 16, dtype=np.float32)
        assert lib.sgx_build_obs_lut(C.byref(cfg), lut.ctypes.data_as(C.POINTER(C.c_float))) == 0
        lut = lut.reshape(67, 16)
        mids = np.asarray(ref[name]['p_obs_mids'], dtype=np.float32)
        ranges = np.asarray(ref[name]['p_obs_ranges'], dtype=np.float32)
        for ch in range(67):
            for i in range(16):
                if ch < 38:
                    t = ch + 1 if ch < 12 else (ch - 11 if ch < 25 else ch - 24)
                    raw = np.float32(1.0 if i == t else 0.0)
                elif ch in (39, 40):
                    raw = np.float32(i - 3)
                else:
                    raw = np.float32(i)
                want = (raw - mids[ch]) / ranges[ch]
                assert want.dtype == np.float32
                assert lut[ch, i].tobytes() == np.float32(want).tobytes(), (name, ch, i)
    # the bit patterns SURVEY A.6 lists
    cfg = _lib.make_config(VARIANTS['standard'])
    lut = np.zeros(67 * 16, dtype=np.float32)
    lib.sgx_build_obs_lut(C.byref(cfg), lut.ctypes.data_as(C.POINTER(C.c_float)))
    lut = lut.reshape(67, 16)
    assert lut[41 + 2, 1].view(np.uint32) == 0xbf19999a      # miner (hi 5), count 1 -> -0.6
    assert lut[41 + 6, 1].view(np.uint32) == 0xbeaaaaab      # major (hi 3), count 1 -> -0.33333334
    assert lut[41 + 11, 1].view(np.uint32) == 0xbf2aaaab     # bomb (hi 6), count 1 -> -0.6666667
    assert lut[41 + 0, 1].view(np.uint32) == 0xbf400000      # spy (hi 8), count 1 -> -0.75


def test_full_obs_lut_bit_exact_vs_reference_constants(lib):
    ref = load_variants_json()['variants']
    for name, v in VARIANTS.items():
        cfg = _lib.make_config(v)
        lut = np.zeros(79 * 16, dtype=np.float32)
        assert lib.sgx_build_full_obs_lut(C.byref(cfg), lut.ctypes.data_as(C.POINTER(C.c_float))) == 0
        lut = lut.reshape(79, 16)
        mids = np.asarray(ref[name]['f_obs_mids'], dtype=np.float32)
        ranges = np.asarray(ref[name]['f_obs_ranges'], dtype=np.float32)
        for ch in range(79):
            for i in range(16):
                if ch < 50:
                    t = (ch % 12 + 1) if ch < 24 else ((ch - 24) % 13 + 1)
                    raw = np.float32(1.0 if i == t else 0.0)
                elif ch in (51, 52):
                    raw = np.float32(i - 3)
                else:
                    raw = np.float32(i)
                want = (raw - mids[ch]) / ranges[ch]
                assert lut[ch, i].tobytes() == np.float32(want).tobytes(), (name, ch, i)


def test_original_channel_luts_bit_exact_vs_reference_constants(lib):
    """obs_channel_mode='original': LUT[ch][v] = (float32(v) - mid) / range with the reference's constants
    (tests/golden/orig_norm.json, recorded from maenv:87-199, 388-396)."""
    import json
    with open(os.path.join(ROOT, 'tests', 'golden', 'orig_norm.json')) as f:
        ref = json.load(f)
    for name, r in ref.items():
        cfg = _lib.make_config(VARIANTS[name])
        for full, nch, rec, key in ((0, 32, (4, 5), 'p_obs'), (1, 33, (3, 4), 'f_obs')):
            lut = np.zeros(nch * 16, dtype=np.float32)
            assert lib.sgx_build_original_obs_lut(C.byref(cfg), full, lut.ctypes.data_as(C.POINTER(C.c_float))) == 0
            lut = lut.reshape(nch, 16)
            mids = np.asarray(r[key + '_mids'], dtype=np.float32)
            ranges = np.asarray(r[key + '_ranges'], dtype=np.float32)
            for ch in range(nch):
                for i in range(16):
                    raw = np.float32(i - 3) if ch in rec else np.float32(i)
                    want = (raw - mids[ch]) / ranges[ch]
                    assert lut[ch, i].tobytes() == np.float32(want).tobytes(), (name, full, ch, i)


def test_facade_norm_constants_match_reference():
    """stratego_env_amd.obs_norm (the facade's _p_obs_mids / _f_obs_ranges ... attributes) vs the reference dump."""
    import json
    from stratego_env_amd import obs_norm
    ext = load_variants_json()['variants']
    with open(os.path.join(ROOT, 'tests', 'golden', 'orig_norm.json')) as f:
        orig = json.load(f)
    for ref, original in ((ext, False), (orig, True)):
        for name, r in ref.items():
            for full, key in ((False, 'p_obs'), (True, 'f_obs')):
                hi, lo = obs_norm.obs_highs_lows(VARIANTS[name].piece_counts, full, original)
                rg, md = obs_norm.ranges_mids(hi, lo)
                assert rg.dtype == np.float32 and rg.shape == (1, 1, hi.shape[0])
                assert np.array_equal(md.reshape(-1), np.asarray(r[key + '_mids'], dtype=np.float32)), (name, key)
                assert np.array_equal(rg.reshape(-1), np.asarray(r[key + '_ranges'], dtype=np.float32)), (name, key)


def test_bad_config_rejected(lib):
    cfg = _lib.make_config(VARIANTS['barrage'])
    cfg.rows = 2
    lut = (C.c_float * (67 * 16))()
    assert lib.sgx_build_obs_lut(C.byref(cfg), lut) == -1
    assert b'at least 3' in lib.sgx_last_error()


def test_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _lib.make_config(VARIANTS['barrage'])
    h = C.c_void_p()
    rc = lib.sgx_create(C.byref(cfg), 16, 0, 1, 0, C.byref(h))
    assert rc != 0 and not h.value
    assert lib.sgx_last_error()
    from stratego_env_amd.vec_env import VecStrategoEnv
    with pytest.raises(_lib.SgxError):
        VecStrategoEnv('barrage', 4)


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under stratego_env_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'stratego_env_amd')
    for dp, dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                # (the device sources may NAME the oracle file in a comment -- the RNG is restated there -- nothing else may)
                assert 'stratego_oracle' not in src or os.path.basename(dp) == 'csrc', f


def test_hot_kernels_use_no_scratch_memory(tmp_path):
    """Resource guard for the kernels bench.py measures: the gfx950 ISA of step_kernel<R,C,0,false> (partial observation,
    perspective mask) must not spill (scratch = 0) and must fit the register budget of its occupancy target.  (A helper that
    is only reachable through a rare flag once cost the hot kernel 30 VGPRs and scratch: such paths get their own instantiation.)"""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    out = tmp_path / 'k.s'
    subprocess.check_call([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-I', hip_build.INCLUDE,
                           hip_build.SRC, '-o', str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    seen = 0
    for m in re.finditer(r'\.name:\s+(\S*step_kernelILi(\d+)ELi(\d+)ELi0ELb0E\S*)', text):
        blk = text[max(0, m.start() - 1500):m.end() + 800]
        scratch = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
        vgpr = int(re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1))
        cells = int(m.group(2)) * int(m.group(3))
        assert scratch == 0, (m.group(1), scratch)
        assert vgpr <= (64 if cells <= 64 else 80), (m.group(1), vgpr)      # 8 waves/SIMD on small boards, 6 on the others
        seen += 1
    assert seen == 7


def test_missing_library_fails_loudly():
    """No fallback path: a missing .so is an error at load time, and the env class propagates it."""
    with pytest.raises(_lib.SgxError) as ei:
        _lib.load('/nonexistent/libstratego_mi355x.so')
    assert 'no CPU fallback' in str(ei.value)
