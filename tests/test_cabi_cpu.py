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
    assert lib.sgx_abi_version() == _lib.ABI_VERSION == 15


def test_struct_sizes_match_header():
    # sgx_config: 4 + 12 + 1 int32 + SGX_MAX_CELLS bytes; sgx_step_io: 12 pointers + 2 int32
    assert C.sizeof(_lib.SgxConfig) == 17 * 4 + 1024
    assert C.sizeof(_lib.SgxStepIO) == 12 * 8 + 8
    assert C.sizeof(_lib.SgxTrajIO) == (12 * 8 + 8) + 2 * 4 + 8 + 8          # sgx_traj_io: the step io, n_slots, results_per_slot, slot_envs, actions_log_dev
    assert _lib.SgxTrajIO.slot_envs.offset == 112 and _lib.SgxTrajIO.actions_log_dev.offset == 120
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


def _kernel_resources(tmp_path):
    """{mangled kernel name: {vgpr, sgpr, scratch, lds}} read from the gfx950 code object inside the SHIPPED library (the notes of the
    offload bundle: what the GPU box will run, and seconds instead of the minutes of a fresh hipcc -S)."""
    import shutil
    import subprocess
    llvm = '/opt/rocm/lib/llvm/bin'
    so = tmp_path / 'lib.so'
    hip_build.build()                                  # (a no-op when the library matches the sources: its build id is their hash)
    shutil.copy(hip_build.LIB_PATH, so)
    subprocess.check_call([os.path.join(llvm, 'llvm-objdump'), '--offloading', str(so)], stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    cos = [f for f in os.listdir(tmp_path) if 'gfx950' in f]
    assert len(cos) == 1, cos
    notes = subprocess.run([os.path.join(llvm, 'llvm-readelf'), '--notes', str(tmp_path / cos[0])], capture_output=True, text=True, check=True).stdout
    res = {}
    for blk in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
        res[re.search(r'\.name:\s+(\S+)', blk).group(1)] = dict(vgpr=g('vgpr_count'), sgpr=g('sgpr_count'), scratch=g('private_segment_fixed_size'),
                                                                 lds=g('group_segment_fixed_size'))
    return res


def test_hot_kernels_use_no_scratch_memory(tmp_path):
    """Resource guard for the kernels bench.py measures, read from the shipped binary: step_kernel<R,C,0,false> (partial observation,
    perspective mask) must not spill (scratch = 0) and must fit the register budget of its occupancy target (a helper that is only
    reachable through a rare flag once cost the hot kernel 30 VGPRs and scratch: such paths get their own instantiation); the multi-step
    kernel of the headline boards (steps_kernel<10,10,*>, <15,15,0>) must not spill either, and the smaller boards' stay within the few
    dwords their 64-register budget costs them."""
    res = _kernel_resources(tmp_path)
    seen = 0
    for name, r in res.items():
        m = re.search(r'11step_kernelILi(\d+)ELi(\d+)ELi0ELb0E', name)
        if not m:
            continue
        cells = int(m.group(1)) * int(m.group(2))
        assert r['scratch'] == 0, (name, r)
        assert r['vgpr'] <= (64 if cells <= 64 else 80), (name, r)      # 8 waves/SIMD on small boards, 6 on the others
        seen += 1
    assert seen == 7
    seen = 0
    for name, r in res.items():
        m = re.search(r'12steps_kernelILi(\d+)ELi(\d+)ELi(\d+)E', name)
        if not m:
            continue
        cells, kind = int(m.group(1)) * int(m.group(2)), int(m.group(3))
        if cells >= 36 and cells % 4 == 0 and kind in (0, 8):           # 6x6, 8x8, 10x10 (15x15 below): no scratch in the steps loop
            assert r['scratch'] == 0, (name, r)
        if cells >= 100:
            assert r['scratch'] == 0, (name, r)
        assert r['scratch'] <= 128, (name, r)
        seen += 1
    assert seen >= 7


def test_steps_kernel_reads_its_parameters_with_scalar_loads(tmp_path):
    """The multi-step kernel re-reads its parameter block every step.  Through a generic pointer those re-reads were vector loads
    (flat_load + v_readfirstlane) that retire in order with the wave's outstanding observation stores; through the kernel-argument
    address space (SGX_KERNARG) they are scalar loads.  Guard: no flat_load in the shipped steps_kernel<10,10,0>."""
    import subprocess
    _kernel_resources(tmp_path)                      # (unbundles the gfx950 code object into tmp_path)
    co = [f for f in os.listdir(tmp_path) if 'gfx950' in f][0]
    sym = '_ZN12_GLOBAL__N_112steps_kernelILi10ELi10ELi0ELi0EEEvNS_15WaveStepsParamsE'
    asm = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--mcpu=gfx950', '--disassemble-symbols=' + sym, str(tmp_path / co)],
                         capture_output=True, text=True, check=True).stdout
    assert asm.count('s_load_dword') > 40 and 's_endpgm' in asm
    assert 'flat_load' not in asm and 'scratch_' not in asm


def test_missing_library_fails_loudly():
    """No fallback path: a missing .so is an error at load time, and the env class propagates it."""
    with pytest.raises(_lib.SgxError) as ei:
        _lib.load('/nonexistent/libstratego_mi355x.so')
    assert 'no CPU fallback' in str(ei.value)
