"""ctypes binding of libstratego_mi355x.so (declared in include/stratego_mi355x.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the env classes
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

from .build import LIB_PATH, BUILTIN_GEOMETRIES, build_geometry

SGX_MAX_CELLS = 1024
SGX_OBS_LUT_STRIDE = 16
PO_OBS_CHANNELS = 67
FO_OBS_CHANNELS = 79
PO_OBS_CHANNELS_ORIGINAL = 32
FO_OBS_CHANNELS_ORIGINAL = 33
ABI_VERSION = 15
STEP_ACTIONS_1D, STEP_ALLOW_OSCILLATION, STEP_RAW_OBS, STEP_ACTIONS_POSITIONS, STEP_ORIGINAL_CHANNELS = 1, 2, 4, 8, 16
STEP_MASK_1D, STEP_MASK_STATE_COORDS = 32, 64
STEP_COMPACT_OBS, STEP_COMPACT_MASK = 128, 256
OUT_FULL_OBS, OUT_MAX_TRIALS = 1024, 64
LAUNCH_WAVE, LAUNCH_LANE, LAUNCH_MULTI_STEP, LAUNCH_MULTI_STEP_WAVE = 0, 1, 2, 3

# every symbol include/stratego_mi355x.h declares
EXPORTED_SYMBOLS = (
    'sgx_abi_version', 'sgx_build_id', 'sgx_supports_geometry', 'sgx_last_error', 'sgx_num_envs', 'sgx_record_bytes', 'sgx_spatial_channels', 'sgx_num_spatial_actions',
    'sgx_action_size_1d', 'sgx_build_obs_lut', 'sgx_build_full_obs_lut', 'sgx_build_original_obs_lut', 'sgx_create', 'sgx_destroy', 'sgx_set_nt_stores', 'sgx_set_lane_kernel', 'sgx_set_half_wave', 'sgx_set_steps_barrier', 'sgx_set_multi_step', 'sgx_last_launch_kind', 'sgx_set_xcd_skew', 'sgx_set_xcd_shares', 'sgx_get_xcd_shares', 'sgx_set_setup_table', 'sgx_reset',
    'sgx_observe', 'sgx_time_observe', 'sgx_mem_probe', 'sgx_store_probe', 'sgx_alloc_outputs', 'sgx_set_placement_target', 'sgx_free_outputs', 'sgx_step', 'sgx_host_alloc', 'sgx_host_free', 'sgx_step_sync', 'sgx_step_n', 'sgx_step_ring', 'sgx_step_traj', 'sgx_rollout', 'sgx_compact_obs_stride', 'sgx_compact_mask_words', 'sgx_decode_obs', 'sgx_decode_mask', 'sgx_sample_valid', 'sgx_choose_actions', 'sgx_export_state', 'sgx_import_state', 'sgx_import_state_checked', 'sgx_step_states', 'sgx_set_general_states', 'sgx_copy_envs', 'sgx_expand', 'sgx_get_env_info',
)


class SgxConfig(C.Structure):
    _fields_ = [('rows', C.c_int32), ('cols', C.c_int32), ('max_turns', C.c_int32), ('usable_rows', C.c_int32),
                ('piece_counts', C.c_int32 * 12), ('capture_capacity', C.c_int32), ('obstacles', C.c_uint8 * SGX_MAX_CELLS)]


class SgxStepIO(C.Structure):
    _fields_ = [('actions_dev', C.c_void_p), ('obs_dev', C.c_void_p), ('fobs_dev', C.c_void_p), ('mask_dev', C.c_void_p),
                ('reward_dev', C.c_void_p), ('done_dev', C.c_void_p), ('player_dev', C.c_void_p),
                ('invalid_action_dev', C.c_void_p), ('ending_invalid_dev', C.c_void_p), ('final_obs_dev', C.c_void_p),
                ('final_fobs_dev', C.c_void_p), ('next_actions_dev', C.c_void_p), ('auto_reset', C.c_int32),
                ('flags', C.c_int32)]


class SgxTrajIO(C.Structure):
    _fields_ = [('io', SgxStepIO), ('n_slots', C.c_int32), ('results_per_slot', C.c_int32), ('slot_envs', C.c_int64), ('actions_log_dev', C.c_void_p)]


class SgxOutputs(C.Structure):
    _fields_ = [('obs_dev', C.c_void_p), ('fobs_dev', C.c_void_p), ('mask_dev', C.c_void_p),
                ('obs_bytes', C.c_int64), ('fobs_bytes', C.c_int64), ('mask_bytes', C.c_int64), ('peak_extra_bytes', C.c_int64),
                ('n_trials', C.c_int32), ('n_ftrials', C.c_int32),
                ('trial_us', C.c_float * OUT_MAX_TRIALS), ('ftrial_us', C.c_float * OUT_MAX_TRIALS),
                ('device', C.c_int32), ('reserved_', C.c_int32)]


class SgxError(RuntimeError):
    pass


_libs = {}


def _bind(L):
    vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
    L.sgx_abi_version.restype = C.c_int
    L.sgx_abi_version.argtypes = []
    L.sgx_build_id.restype = C.c_char_p
    L.sgx_build_id.argtypes = []
    L.sgx_supports_geometry.restype = C.c_int
    L.sgx_supports_geometry.argtypes = [C.c_int32, C.c_int32]
    L.sgx_last_error.restype = C.c_char_p
    L.sgx_last_error.argtypes = []
    L.sgx_num_envs.restype = i64
    L.sgx_num_envs.argtypes = [vp]
    L.sgx_record_bytes.restype = i64
    L.sgx_record_bytes.argtypes = [vp]
    L.sgx_spatial_channels.restype = C.c_int
    L.sgx_spatial_channels.argtypes = [vp]
    L.sgx_num_spatial_actions.restype = i64
    L.sgx_num_spatial_actions.argtypes = [vp]
    L.sgx_action_size_1d.restype = i64
    L.sgx_action_size_1d.argtypes = [vp]
    L.sgx_build_obs_lut.restype = C.c_int
    L.sgx_build_obs_lut.argtypes = [C.POINTER(SgxConfig), C.POINTER(C.c_float)]
    L.sgx_build_full_obs_lut.restype = C.c_int
    L.sgx_build_full_obs_lut.argtypes = [C.POINTER(SgxConfig), C.POINTER(C.c_float)]
    L.sgx_build_original_obs_lut.restype = C.c_int
    L.sgx_build_original_obs_lut.argtypes = [C.POINTER(SgxConfig), C.c_int32, C.POINTER(C.c_float)]
    L.sgx_create.restype = C.c_int
    L.sgx_create.argtypes = [C.POINTER(SgxConfig), i64, C.c_int, u64, i64, C.POINTER(vp)]
    L.sgx_destroy.restype = C.c_int
    L.sgx_destroy.argtypes = [vp]
    L.sgx_set_nt_stores.restype = C.c_int
    L.sgx_set_nt_stores.argtypes = [vp, C.c_int32]
    L.sgx_set_lane_kernel.restype = C.c_int
    L.sgx_set_lane_kernel.argtypes = [vp, C.c_int32]
    L.sgx_set_half_wave.restype = C.c_int
    L.sgx_set_half_wave.argtypes = [vp, C.c_int32]
    L.sgx_set_steps_barrier.restype = C.c_int
    L.sgx_set_steps_barrier.argtypes = [vp, C.c_int32]
    L.sgx_set_multi_step.restype = C.c_int
    L.sgx_set_multi_step.argtypes = [vp, C.c_int32]
    L.sgx_last_launch_kind.restype = C.c_int
    L.sgx_last_launch_kind.argtypes = [vp]
    L.sgx_set_xcd_skew.restype = C.c_int
    L.sgx_set_xcd_skew.argtypes = [vp, C.c_int32]
    L.sgx_set_xcd_shares.restype = C.c_int
    L.sgx_set_xcd_shares.argtypes = [vp, C.POINTER(C.c_int32)]
    L.sgx_get_xcd_shares.restype = C.c_int
    L.sgx_get_xcd_shares.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.sgx_set_setup_table.restype = C.c_int
    L.sgx_set_setup_table.argtypes = [vp, vp, i64]
    L.sgx_reset.restype = C.c_int
    L.sgx_reset.argtypes = [vp, vp, vp, vp, vp]
    L.sgx_observe.restype = C.c_int
    L.sgx_observe.argtypes = [vp, vp, vp, vp, vp, C.c_int32, vp]
    L.sgx_time_observe.restype = C.c_int
    L.sgx_time_observe.argtypes = [vp, vp, vp, C.c_int32, vp, C.POINTER(C.c_float)]
    L.sgx_mem_probe.restype = C.c_int
    L.sgx_mem_probe.argtypes = [C.c_int, vp, i64, C.c_int32, vp, C.POINTER(C.c_float)]
    L.sgx_alloc_outputs.restype = C.c_int
    L.sgx_alloc_outputs.argtypes = [vp, C.c_int32, i64, C.c_int32, vp, C.POINTER(SgxOutputs)]
    L.sgx_set_placement_target.restype = C.c_int
    L.sgx_set_placement_target.argtypes = [vp, C.c_float]
    L.sgx_free_outputs.restype = C.c_int
    L.sgx_free_outputs.argtypes = [vp, C.POINTER(SgxOutputs)]
    L.sgx_step.restype = C.c_int
    L.sgx_step.argtypes = [vp, C.POINTER(SgxStepIO), vp]
    L.sgx_host_alloc.restype = C.c_int
    L.sgx_host_alloc.argtypes = [vp, i64, C.POINTER(vp), C.POINTER(vp)]
    L.sgx_host_free.restype = C.c_int
    L.sgx_host_free.argtypes = [vp, vp]
    L.sgx_step_sync.restype = C.c_int
    L.sgx_step_sync.argtypes = [vp, C.POINTER(SgxStepIO), vp]
    L.sgx_step_n.restype = C.c_int
    L.sgx_step_n.argtypes = [vp, C.POINTER(SgxStepIO), C.c_int32, vp]
    L.sgx_step_ring.restype = C.c_int
    L.sgx_step_ring.argtypes = [vp, C.POINTER(SgxStepIO), C.c_int32, C.c_int32, C.c_int32, vp]
    L.sgx_step_traj.restype = C.c_int
    L.sgx_step_traj.argtypes = [vp, C.POINTER(SgxTrajIO), C.c_int32, C.c_int32, vp]
    L.sgx_store_probe.restype = C.c_int
    L.sgx_store_probe.argtypes = [C.c_int, vp, i64] + [C.c_int32] * 10 + [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.sgx_rollout.restype = C.c_int
    L.sgx_rollout.argtypes = [vp, C.POINTER(SgxStepIO), C.c_int32, C.c_int32, vp]
    L.sgx_compact_obs_stride.restype = i64
    L.sgx_compact_obs_stride.argtypes = [vp]
    L.sgx_compact_mask_words.restype = i64
    L.sgx_compact_mask_words.argtypes = [vp]
    L.sgx_decode_obs.restype = C.c_int
    L.sgx_decode_obs.argtypes = [vp, vp, vp, vp]
    L.sgx_decode_mask.restype = C.c_int
    L.sgx_decode_mask.argtypes = [vp, vp, vp, vp]
    L.sgx_sample_valid.restype = C.c_int
    L.sgx_sample_valid.argtypes = [vp, vp, vp, vp]
    L.sgx_choose_actions.restype = C.c_int
    L.sgx_choose_actions.argtypes = [vp, vp, vp, C.c_float, C.c_int32, vp, vp]
    L.sgx_export_state.restype = C.c_int
    L.sgx_export_state.argtypes = [vp, vp, vp, vp]
    L.sgx_import_state.restype = C.c_int
    L.sgx_import_state.argtypes = [vp, vp, vp, vp]
    L.sgx_import_state_checked.restype = C.c_int
    L.sgx_import_state_checked.argtypes = [vp, vp, vp, vp, vp]
    L.sgx_step_states.restype = C.c_int
    L.sgx_step_states.argtypes = [vp, vp, vp, vp, C.POINTER(SgxStepIO), vp, vp, C.c_int32, vp]
    L.sgx_set_general_states.restype = C.c_int
    L.sgx_set_general_states.argtypes = [vp, C.c_int32]
    L.sgx_copy_envs.restype = C.c_int
    L.sgx_copy_envs.argtypes = [vp, vp, vp, vp, i64, vp]
    L.sgx_expand.restype = C.c_int
    L.sgx_expand.argtypes = [vp, vp, vp, C.POINTER(SgxStepIO), vp]
    L.sgx_get_env_info.restype = C.c_int
    L.sgx_get_env_info.argtypes = [vp, vp, vp]
    return L


def load(path=None):
    """Load the HIP library; raises if it has not been built (no fallback path exists).

    `path` / $SGX_LIB_PATH select an alternative build of the same library (kernel experiments only)."""
    path = path or os.environ.get('SGX_LIB_PATH', LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise SgxError("libstratego_mi355x.so is not built (%s missing). Run `python -m stratego_env_amd.build` "
                       "or __graft_entry__.build(); there is no CPU fallback." % path)
    L = C.CDLL(path)
    missing = [sym for sym in EXPORTED_SYMBOLS if not hasattr(L, sym)]
    if missing:
        raise SgxError("%s does not export %s: it was built from other sources; rebuild with `python -m stratego_env_amd.build`"
                       % (path, ', '.join(missing)))
    L = _bind(L)
    if L.sgx_abi_version() != ABI_VERSION:       # struct layouts differ between ABI versions: never bind a stale library
        raise SgxError("%s has ABI version %d, this package needs %d; rebuild with `python -m stratego_env_amd.build`"
                       % (path, L.sgx_abi_version(), ABI_VERSION))
    # the binary must come from the sources next to it: a stale library of the same ABI version would load and run -- and every parity
    # claim of the test suite would be about another program.  $SGX_ALLOW_FOREIGN_BUILD=1 lets a kernel experiment built with other
    # flags or sources through (tools/ A/B runs), loudly.
    from . import build as _build
    have, want = L.sgx_build_id().decode('ascii', 'replace'), _build.source_hash()
    if have != want:
        msg = ("%s was built from other sources (build id %s, the sources here hash to %s); rebuild with "
               "`python -m stratego_env_amd.build`" % (path, have, want))
        if os.environ.get('SGX_ALLOW_FOREIGN_BUILD') != '1':
            raise SgxError(msg)
        import warnings
        warnings.warn(msg + " -- loaded anyway because SGX_ALLOW_FOREIGN_BUILD=1", RuntimeWarning)
    L.build_id = have
    _libs[path] = L
    return L


def load_for_geometry(rows, columns, path=None):
    """The library that holds the kernels of a rows x columns board: the main one for the sizes of the reference's variants, else
    a library of its own built on first use (build.build_geometry).  An explicit `path` / $SGX_LIB_PATH wins."""
    if path or os.environ.get('SGX_LIB_PATH') or (int(rows), int(columns)) in BUILTIN_GEOMETRIES:
        return load(path)
    return load(build_geometry(rows, columns))


def check(rc, lib=None):
    if rc != 0:
        raise SgxError("libstratego_mi355x error %d: %s" % (rc, (lib or load()).sgx_last_error().decode('utf-8', 'replace')))


def make_config(variant) -> SgxConfig:
    """stratego_env_amd.config.Variant -> sgx_config."""
    cfg = SgxConfig()
    cfg.rows, cfg.cols = variant.rows, variant.columns
    cfg.max_turns = variant.max_turns
    cfg.usable_rows = variant.initial_state_usable_rows
    for i, n in enumerate(variant.piece_counts):
        cfg.piece_counts[i] = n
    cfg.capture_capacity = int(getattr(variant, 'capture_capacity', 0))
    for r, c in variant.obstacle_locations:
        cfg.obstacles[r * variant.columns + c] = 1
    return cfg
